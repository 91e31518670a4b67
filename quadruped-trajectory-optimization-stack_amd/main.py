"""Command-line twin of the reference's ``./main``: same argv, same CSV, same exit status.

``args['scripts']['run']`` of the reference (QTOS/utils.py:17, 'docker exec <id> ./main') can be
pointed at ``python -m qtos_amd.main --out build/traj.csv`` unchanged otherwise.
"""
import sys

from . import flags
from .planner import LocalPlanner, TOWR_HEIGHTFIELD


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    out, hf = "build/traj.csv", None
    for opt in ("--out", "--heightfield"):
        if opt in argv:
            i = argv.index(opt)
            val = argv[i + 1]
            del argv[i:i + 2]
            if opt == "--out":
                out = val
            else:
                hf = val
    args = flags.parse_flags(argv)
    lp = LocalPlanner(max_batch=1)
    try:
        import os
        path = hf or TOWR_HEIGHTFIELD
        if os.path.exists(path):
            lp.load_heightfield_file(path, args.get('-resolution'))
        status = lp.solve(args, out_csv=out)
        print("status -> %d" % status)  # the line the reference's log carries (logs/towr_log.out:85)
        return status
    finally:
        lp.close()


if __name__ == "__main__":
    sys.exit(main())
