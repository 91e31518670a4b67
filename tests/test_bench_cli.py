"""bench.py's command line on the CPU: the defaults the driver relies on, and every bench command README.md prints parses
(the eight-GPU commands of BASELINE configs[3] / configs[4] are the driver's to run; here they must at least be commands)."""
import os
import re
import shlex
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_defaults_are_the_headline_run():
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.batch) == (1, 100, 5, 256)
    # the line's metric / value / roofline are the TROT's (the gait BASELINE.json's metric names); the walk of the reference's golden
    # plans is the second leg of the same command, timed with the same --steps
    assert (a.transcription, a.gait, a.workload) == ("knots100", "trot", "exp1_flat")
    assert not a.no_second_gait and not hasattr(a, "trot_steps")
    assert a.settle == 50 and a.settle_tol == 0.02 and not a.no_pattern     # adaptive warm-up on, launch pattern on
    assert a.events_every == 4     # per-kernel HIP events on every fourth timed step (they cost a batch 1.9 %)
    # the terrain workloads and the receding windows keep the walk (their goals, schedules and parity tests are the walk's)
    assert bench.parse_args(["--workload", "exp5_step"]).gait == "walk"
    assert bench.parse_args(["--workload", "mixed"]).gait == "walk"
    assert bench.parse_args(["--transcription", "knots200", "--workload", "mpc_random"]).gait == "walk"
    assert bench.parse_args(["--gait", "walk"]).gait == "walk" and bench.parse_args(["--no-trot"]).no_second_gait
    assert not a.full_system and not a.full_swings and not a.plain_mu and not a.force_torchrun
    assert a.cpu_sample > 0 and not a.no_parity


def test_readme_bench_commands_parse():
    import bench
    text = open(os.path.join(ROOT, "README.md")).read()
    cmds = [l.strip() for l in text.splitlines() if l.startswith("    python ") and "bench.py" in l]
    assert len(cmds) >= 3
    seen = set()
    for c in cmds:
        argv = shlex.split(c.split("#")[0])
        argv = argv[argv.index("bench.py") + 1:]
        a = bench.parse_args(argv)
        seen.add((a.gpus, a.workload, a.transcription))
        if "torch.distributed.run" in c:   # the launcher's process count is the --gpus the ranks are told
            assert re.search(r"--nproc-per-node (\d+)", c).group(1) == str(a.gpus) and "--master-addr 127.0.0.1" in c
    # BASELINE configs[3]: batch 2048 mixed terrains over 8 GPUs; configs[4]: 200-knot receding windows over 8 GPUs
    assert (8, "mixed", "knots100") in seen and (8, "mpc_random", "knots200") in seen
