// qtos_planner.hip -- C ABI (include/qtos_planner.h) over the gfx950 kernels in kernels.hpp.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared qtos_planner.hip -o libqtos_planner.so
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <map>
#include <vector>

#include "kernels.hpp"
#include "kkt2.hpp"
#include "kkt3.hpp"
#include "kkt5.hpp"
#include "csv_writer.hpp"

using namespace qtos;

#define HIPCHK(p, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (p)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return e_ == hipErrorOutOfMemory ? -3 : -2;                                             \
    }                                                                                         \
  } while (0)

struct QtosPlanner {
  HostModel M;
  Symbolic S;
  DevPlan dp;
  SamplePlan sp;
  int device = 0, max_batch = 0;
  std::vector<void *> allocs;  // everything to free
  // workspace
  DevWork wk;
  double *d_start = nullptr, *d_goal = nullptr, *d_nodes = nullptr, *d_warm = nullptr;
  int *d_map = nullptr;
  double *d_height = nullptr;
  hipStream_t own_stream = nullptr;   // stream of the host-pointer entry points (non-blocking: other handles / streams are not synchronised)
  long long *d_totals = nullptr;      // converged problems, iterations: tallied at the end of every plan call (qtos_plan_totals)
  hipStream_t last_stream = nullptr;
  double *d_table = nullptr, *d_tab_dx = nullptr, *d_tab_dy = nullptr;   // nominal-plan table (qtos_set_init_table)
  // A call is served by one or more LANES.  A lane owns everything one host-driven Newton loop needs: the stream its kernels
  // run on (lane 0: the caller's; the others: streams of the planner), a side stream for the chord solve that runs next to a
  // factorisation, the pinned count slots its iterations report to, its events and its two device counters.  A call of more
  // problems than the GPU has compute units is cut into contiguous parts, one per lane: the late iterations of a part's
  // stragglers then run side by side with the other parts' full grids instead of holding the whole call -- the reference's
  // queue of probes (QTOS/generateHeightField.py:375-377) inside ONE call.  The parts never meet: the plans are bit for bit
  // those of a single lane (QTOS_LANES=1).
  struct Lane {
    bool open = false;
    int B = 0, b0 = 0, spec = 0, enq = 0, chk = 0;   // problems, first problem; iterations queued blind / queued in all / whose preceding counts have been read
    int n_informed = 0;                      // launches of the call that waited for the host to read the counts
    bool by_pattern = false;                 // the blind slots of this call follow the handle's launch pattern (counts behind the last one only)
    unsigned spins = 0;                      // polls that found no counts yet
    hipStream_t st = nullptr;                // where this lane's kernels go in the call in flight
    hipStream_t own = nullptr, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr;
    DevWork W;
    int *h_active = nullptr;      // pinned + mapped, four words per launch slot: {unfinished problems behind it, of those flagged for a chord step, sequence number}
                                  // packed into the first two, then the code of the earliest slot a problem sat out and the slots that had work
    int *h_active_dev = nullptr;  // the same memory as the device sees it (k_post_counts stores there: no copy engine between two kernels)
    int *d_n_active = nullptr;    // the lane's four device counters (DevWork::n_active)
    std::vector<hipEvent_t> ev;   // 5 per iteration (kkt begin / end, chord begin / end, counts posted) + 3
    std::vector<char> was_kkt, was_chord;   // per iteration of the last call: which solve kernels were launched
    int last_launches = 0, last_iters = 0;
  };
  std::vector<Lane> lanes;
  int lanes_used = 1, lane_chunk = 256;      // lanes of the call in flight; problems per lane above which a call is cut (the GPU's compute units)
  bool call_open = false;
  unsigned seq = 0;                          // sequence number of the call in flight, stamped into the count words (k_post_counts)
  hipStream_t call_stream = nullptr;
  hipEvent_t ev_in = nullptr;                // the caller's stream at submit time: the other lanes start behind it
  unsigned call_seq = 0;             // sequence number of the last call submitted
  bool use_kkt3 = false;             // k_kkt3 (kkt3.hpp) instead of k_kkt2: chosen by qtos_planner_create
  bool use_kkt5 = false;             // k_kkt5 (kkt5.hpp): two stages per set of barriers, Symbolic::pair_mode
  QtosEnv env;                       // the environment as qtos_planner_create found it (env.hpp; qtos_env reports it)
  int spec_next = 1;                 // blind iterations of the next call: the iterations the last one took
  int spec_cap = 1;                  // limit of the blind iterations (qtos_set_speculation): 1 = off
  // Launch pattern (round 6): the solve kernels every launch slot of the handle's last two calls needed (bit 0 the
  // factorising kernel, bit 1 k_chord), as far as the two agree -- the walk's and the trot's batches take kkt, kkt, kkt, chord
  // every time.  qtos_plan_submit queues that prefix at once, without a look at the counts; the counts behind its last slot
  // say whether anything is left (qtos_plan_poll goes on informed) and whether a problem found the wrong kernel (it sat the
  // launch out, kernels.hpp k_step: the pattern is cut in front of that slot).
  bool per_kernel_events = true;     // HIP events around every solve kernel and behind every slot (qtos_last_timing*); qtos_set_kernel_events(0): only the call's first and last
  bool spec_pattern = true;
  std::vector<char> pat, obs_prev;   // the prefix queued blind by the next call; the kinds the last call's slots ran
  int max_slots = 0;                 // launch slots a call may use (iterations + launches a problem sat out)
  long long n_pattern_calls = 0, n_pattern_misses = 0;
  std::atomic<int> busy{0};
  size_t kkt_lds = 0, eval_lds = 0;
  void (*chord_fn)(DevPlan, DevWork, int) = nullptr; // k_chord instantiated for this front size (null: chord steps off)
  void (*kkt_fn)(DevPlan, DevWork, int) = nullptr;   // k_kkt / k_kkt2 instantiated for this front size
  int kkt_threads = KT;
  std::string err;

  template <class T>
  int upload(const std::vector<T> &v, const T **out) {
    T *d = nullptr;
    size_t bytes = std::max<size_t>(1, v.size()) * sizeof(T);
    HIPCHK(this, hipMalloc((void **)&d, bytes));
    allocs.push_back(d);
    if (!v.empty()) HIPCHK(this, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return 0;
  }
  template <class T>
  int alloc(T **out, size_t count) {
    T *d = nullptr;
    HIPCHK(this, hipMalloc((void **)&d, std::max<size_t>(1, count) * sizeof(T)));
    HIPCHK(this, hipMemset(d, 0, std::max<size_t>(1, count) * sizeof(T)));
    allocs.push_back(d);
    *out = d;
    return 0;
  }
};

// k_kkt is compiled once per front size (multiples of 16 up to 128): the LDS layout and every tile
// loop bound are compile-time constants
// k_kkt2 (16 waves per problem): fronts up to 208 slots
static void (*kkt2_kernel(int F, bool cont, bool kron = false))(DevPlan, DevWork, int) {
#if defined(QTOS_EXPERIMENTS) && !defined(QTOS_DEV_F128)
  if (kron && !cont && F == 128) return k_kkt2<128, false, true>;
#endif   // (the Kronecker assembly: the benchmark's front only)
#define QTOS_KKT2(f) case f: return cont ? k_kkt2<f, true> : k_kkt2<f, false>;
  switch (F) {
#ifdef QTOS_DEV_F128   // (development builds: the benchmark's fronts only, a quarter of the compile time)
    QTOS_KKT2(112) QTOS_KKT2(128)
#else
    QTOS_KKT2(16) QTOS_KKT2(32) QTOS_KKT2(48) QTOS_KKT2(64) QTOS_KKT2(80) QTOS_KKT2(96) QTOS_KKT2(112) QTOS_KKT2(128)
#endif
#if !defined(QTOS_DEV_SMALL) && !defined(QTOS_DEV_F128)
    QTOS_KKT2(144) QTOS_KKT2(160) QTOS_KKT2(176) QTOS_KKT2(192) QTOS_KKT2(208)
#endif
  }
#undef QTOS_KKT2
  return nullptr;
}
// k_kkt3 (k_kkt2's records and arithmetic, the assembly on the waves that idle in phase AB): fronts up to 128 slots
static void (*kkt3_kernel(int F))(DevPlan, DevWork, int) {
#define QTOS_KKT3(f) case f: return k_kkt3<f, 1>;
#ifndef QTOS_DEV_F128
  switch (F) { QTOS_KKT3(16) QTOS_KKT3(32) QTOS_KKT3(48) QTOS_KKT3(64) QTOS_KKT3(80) QTOS_KKT3(96) QTOS_KKT3(112) QTOS_KKT3(128) }
#endif
#undef QTOS_KKT3
  return nullptr;
}
// k_kkt5 (two 16-pivot stages per set of barriers, twelve waves): fronts up to 144 slots
static void (*kkt5_kernel(int F))(DevPlan, DevWork, int) {
#define QTOS_KKT5(f) case f: return k_kkt5<f>;
#ifdef QTOS_DEV_F128
  switch (F) { QTOS_KKT5(112) QTOS_KKT5(128) }
#else
  switch (F) { QTOS_KKT5(96) QTOS_KKT5(112) QTOS_KKT5(128) QTOS_KKT5(144) }
#endif
#undef QTOS_KKT5
  return nullptr;
}
static void (*chord_kernel(int F))(DevPlan, DevWork, int) {
#define QTOS_CHORD(f) case f: return k_chord<f>;
  switch (F) {
#ifdef QTOS_DEV_F128
    QTOS_CHORD(112) QTOS_CHORD(128)
#else
    QTOS_CHORD(16) QTOS_CHORD(32) QTOS_CHORD(48) QTOS_CHORD(64) QTOS_CHORD(80) QTOS_CHORD(96) QTOS_CHORD(112) QTOS_CHORD(128)
#endif
#if !defined(QTOS_DEV_SMALL) && !defined(QTOS_DEV_F128)
    QTOS_CHORD(144) QTOS_CHORD(160) QTOS_CHORD(176) QTOS_CHORD(192) QTOS_CHORD(208)
#endif
  }
#undef QTOS_CHORD
  return nullptr;
}

static int upload_spline(QtosPlanner *p, const Spline &S, SampleSpline *out) {
  std::vector<double> tend(S.n_polys), dur(S.dur);
  double t = 0;
  for (int i = 0; i < S.n_polys; ++i) { t += S.dur[i]; tend[i] = t; }
  std::vector<int> idx;
  for (auto &nd : S.idx)
    for (int i = 0; i < 6; ++i) idx.push_back(nd[i]);
  out->n_polys = S.n_polys;
  int rc;
  if ((rc = p->upload(tend, &out->tend))) return rc;
  if ((rc = p->upload(dur, &out->dur))) return rc;
  if ((rc = p->upload(idx, &out->idx))) return rc;
  return 0;
}

// Which time keys the elimination order uses (model.hpp HostModel::order_rule).  On a reduced base rules 2 and 1 are analysed:
// the smaller front wins, then the fewer stages, then the order that needs no continuation records, then rule 2.  Rule 0 -- the
// order of rounds 1 - 5 -- has the smallest front on some short horizons and is NOT in the automatic choice there: its KKT solve
// loses up to six digits on short trots (model.hpp).  QTOS_ORDER=0 | 1 | 2 forces one.
static int pick_order_rule(const QtosParams &params, const QtosEnv &env) {
  if (env.order >= 0) return env.order;
  // (rules 1 and 2 move the B-spline coefficients of the reduced base: without it -- every base row in the system, the
  //  configuration the internals' tests pin -- the order of rounds 1 - 5 stays)
  if (!params.reduce_base) return 0;
  int best = 2, best_front = 1 << 30, best_stages = 1 << 30, best_cont = 1 << 30;
  for (int rule : {2, 1}) {
    HostModel M;
    Symbolic S;
    S.env = env;
    S.env.debug = 0;
    S.env.dump_first.clear();
    S.cell_mode = 2;
    M.order_rule = rule;
    if (M.build(params) || S.build(M)) continue;
    if (!M.reduce_base) return 0;   // (the model kept the full base: a short horizon or unequal polynomial durations)
    int n_cont = 0;
    for (int k = 0; k < S.n_records; ++k) n_cont += S.srec[S.srec_off[k] + 6];
    const int cont = n_cont > 0;
    if (S.front < best_front || (S.front == best_front && (S.n_stages < best_stages || (S.n_stages == best_stages && cont < best_cont)))) {
      best = rule; best_front = S.front; best_stages = S.n_stages; best_cont = cont;
    }
  }
  if (env.debug) fprintf(stderr, "qtos: elimination order rule %d (front %d, %d stages)\n", best, best_front, best_stages);
  return best;
}

extern "C" {
__global__ void k_debug_eval(DevPlan P, DevWork W, int B);

const char *qtos_last_error(const QtosPlanner *p) { return p ? p->err.c_str() : "null planner"; }

void qtos_planner_destroy(QtosPlanner *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  for (void *a : p->allocs) (void)hipFree(a);
  for (auto &L : p->lanes) {
    for (hipEvent_t e : L.ev) (void)hipEventDestroy(e);
    if (L.h_active) (void)hipHostFree(L.h_active);
    if (L.own) (void)hipStreamDestroy(L.own);
    if (L.side) (void)hipStreamDestroy(L.side);
    for (hipEvent_t e : {L.ev_fork, L.ev_join, L.ev_done})
      if (e) (void)hipEventDestroy(e);
  }
  if (p->ev_in) (void)hipEventDestroy(p->ev_in);
  for (void *q : {(void *)p->d_table, (void *)p->d_tab_dx, (void *)p->d_tab_dy, (void *)p->d_height})
    if (q) (void)hipFree(q);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  if (p->d_totals) (void)hipFree(p->d_totals);
  delete p;
}

// The rows of the inequality blocks for the helper waves of the backward sweep (sweep_backward): a block belongs to the stage
// of its earliest column; rounds of SW_ROUND rows (one helper wave), round i in step i of the chain (stage NS - 1 - i): a
// stage's rounds come behind the step that solves it, in the order the chain meets the stages.  Returns the number of
// rounds (0: the tables cannot be built for this plan).
static int build_sweep_tasks(const HostModel &M, const Symbolic &S, std::vector<SwTask> &tasks, std::vector<int> &cpos, std::vector<int> &c16) {
  const int NS = S.n_stages;
  std::vector<std::vector<SwTask>> by_stage(NS);
  bool ok = true;
  for (const Block &b : M.blocks) {
    if (b.kind != 1) continue;
    int mn = INT_MAX;
    const int c0 = (int)cpos.size();
    for (int a = 0; a < b.n; ++a) {
      const int pos = S.var_pos[M.block_cols[b.col_off + a]];
      if (pos < 0) ok = false;
      cpos.push_back(std::max(pos, 0));
      mn = std::min(mn, pos);
    }
    if (mn < 0 || mn / PIV >= NS) { ok = false; continue; }
    const int k0 = (int)c16.size(), n4 = b.n & ~3;
    auto at = [&](int e) { return (unsigned)cpos[c0 + e] & 0xffffu; };
    for (int q = 0; q < 4; ++q)
      for (int u = 0; u < SW_RU; u += 2) {
        const int e0 = std::max(std::min(q + 4 * u, n4 - 4 + q), 0), e1 = std::max(std::min(q + 4 * (u + 1), n4 - 4 + q), 0);
        c16.push_back((int)(at(std::min(e0, b.n - 1)) | (at(std::min(e1, b.n - 1)) << 16)));
      }
    {
      const int r0 = std::min(n4, b.n - 1), r1 = std::min(n4 + 1, b.n - 1), r2 = std::min(n4 + 2, b.n - 1);
      c16.push_back((int)(at(r0) | (at(r1) << 16)));
      c16.push_back((int)at(r2));
      c16.push_back(0); c16.push_back(0);
    }
    for (int r = 0; r < b.m; ++r) by_stage[mn / PIV].push_back({b.goff + r * b.n, b.n, b.row0 + r, k0, c0, {0, 0, 0}});
  }
  if ((size_t)NS * PIV > 65535) ok = false;
  const SwTask none = {0, 4, -1, 0, 0, {0, 0, 0}};
  int step = 0;
  auto round_of = [&](const SwTask *t, int cnt) {
    for (int i = 0; i < SW_ROUND; ++i) tasks.push_back(i < cnt ? t[i] : none);
    ++step;
  };
  for (int u = NS - 1; u >= 0 && ok; --u) {
    while (step < NS - u) round_of(nullptr, 0);   // (x of stage u is complete behind the barrier of step NS - 1 - u)
    for (size_t i = 0; i < by_stage[u].size(); i += SW_ROUND) round_of(by_stage[u].data() + i, (int)std::min<size_t>(SW_ROUND, by_stage[u].size() - i));
  }
  const int nstep = ((NS + SWD - 1) / SWD) * SWD;
  while (step < nstep) round_of(nullptr, 0);
  if (cpos.empty()) cpos.push_back(0);
  while (c16.size() < 20) c16.push_back(0);
  if (S.env.debug) {
    size_t mx = 0, tot = 0;
    for (auto &v : by_stage) { mx = std::max(mx, v.size()); tot += v.size(); }
    fprintf(stderr, "qtos: sweep ds: %d rounds for %d stages (chain %d steps), %zu rows, most in a stage %zu; rows by stage:", step, NS, nstep, tot, mx);
    for (auto &v : by_stage) fprintf(stderr, " %zu", v.size());
    fprintf(stderr, "\n");
  }
  return ok ? step : 0;
}

int qtos_planner_create(const QtosParams *params, int max_batch, int device, QtosPlanner **out) {
  if (!params || !out || max_batch < 1 || max_batch >= (1 << 24)) return -1;   // (the count words of k_post_counts hold 24 bits per count)
  *out = nullptr;
  QtosPlanner *p = new QtosPlanner();
  p->device = device;
  p->max_batch = max_batch;
  // model + symbolic analysis; if the stage records and cells of the result do not fit the LDS next to the panels, again with
  // smaller records (heavy stages then spill into continuation records): both are rebuilt, the analysis writes into the model
  // Which factor + solve kernel (kkt2.hpp, kkt3.hpp; all give the same plans where they share the arithmetic):
  //   k_kkt2            the default for fronts above 112 slots
  //   k_kkt3, MODE 1    the default up to 112 slots: k_kkt2's records and arithmetic -- bit-identical plans -- with the
  //                     assembly of the records on the waves that have no job in phase AB (-5 % per launch on the trot's
  //                     112 slots, -6 % on reference_compat's 96; +4 % on 128 slots, where seven idle waves are too few;
  //                     profiles/r04_experiments)
  //   k_kkt5            QTOS_KKT=6: two 16-pivot stages per set of barriers (pair-mode analysis)
  // QTOS_KKT=2 / 4 / 6 force k_kkt2 / k_kkt3 / k_kkt5.  (k_kkt3 MODE 0 -- inequality blocks condensed by matrix instructions --
  // and k_kkt4 -- the pipelined stage --, QTOS_KKT=3 / 5 of rounds 4 - 5, were correct and slower: scratch/experiments/, last
  // built from commit 991d29f.)
  p->use_kkt3 = false;
  p->use_kkt5 = false;
  p->env = QtosEnv::parse();   // the ONE place a planner reads the environment
  const QtosEnv &env = p->env;
  p->spec_pattern = env.spec_pattern != 0;
  const int order_rule = pick_order_rule(*params, env);
  {
    int forced = env.kkt;
    if (forced == 3 || forced == 5) { fprintf(stderr, "qtos: QTOS_KKT=%d selected an experiment of rounds 4 - 5 that left the library (scratch/experiments/): default kernel\n", forced); forced = 0; }
    if (forced == 6) {
      // k_kkt5: the analysis in pair mode (one record per pair of stages); applicable without continuation records, with a
      // front of at most 144 slots and everything within the LDS
      p->M = HostModel();
      p->S = Symbolic();
      p->S.env = env;
      if ((p->M.order_rule = order_rule, p->M.build(*params))) { fprintf(stderr, "qtos: %s\n", p->M.err.c_str()); delete p; return -1; }
      p->S.cell_mode = 2;
      p->S.pair_mode = true;
      bool ok = p->S.build(p->M) == 0 && kkt5_kernel(p->S.front) != nullptr && !(p->S.pack_src.size() & 1);
      if (ok) {
        int n_cont = 0;
        for (int r = 0; r < p->S.n_records; ++r) n_cont += p->S.srec[p->S.srec_off[r] + 6];
        ok = n_cont == 0 && kkt5_lds_bytes(p->S.front, p->S.n_stages, p->S.max_srec, p->S.max_drec, p->S.n_cells) <= 160 * 1024 - 256;
      }
      p->use_kkt5 = ok;
      if (env.debug) fprintf(stderr, "qtos: k_kkt5 %s (front %d, %s)\n", ok ? "selected" : "not applicable", p->S.front, p->S.err.c_str());
    }
    if (forced != 2 && !p->use_kkt5) {
      p->M = HostModel();
      p->S = Symbolic();
      p->S.env = env;
      if ((p->M.order_rule = order_rule, p->M.build(*params))) { fprintf(stderr, "qtos: %s\n", p->M.err.c_str()); delete p; return -1; }
      p->S.cell_mode = 2;
      bool ok = p->S.build(p->M) == 0 && p->S.front <= (forced ? 128 : 112) && !(p->S.pack_src.size() & 1);
      if (ok) {
        int n_cont = 0;
        for (int k = 0; k < p->S.n_stages; ++k) n_cont += p->S.srec[p->S.srec_off[k] + 6];
        ok = n_cont == 0 && kkt3_lds_bytes(p->S.front, p->S.n_stages, p->S.max_srec, p->S.max_drec, p->S.n_cells) <= 160 * 1024 - 256;
      }
      p->use_kkt3 = ok;
      if (env.debug) fprintf(stderr, "qtos: k_kkt3 %s (%s)\n", ok ? "selected" : "not applicable", p->S.err.c_str());
    }
  }
#ifdef QTOS_EXPERIMENTS
  bool want_kron = env.kron != 0;   // (experiment: Kronecker assembly of the range-of-motion blocks, k_kkt2<128> only)
#else
  bool want_kron = false;
#endif
  for (int cap : {0, 4096, 3072, 2048}) {
    if (p->use_kkt3 || p->use_kkt5) break;
    p->M = HostModel();
    p->S = Symbolic();
    p->S.env = env;
    if ((p->M.order_rule = order_rule, p->M.build(*params))) { fprintf(stderr, "qtos: %s\n", p->M.err.c_str()); delete p; return -1; }
    p->S.cell_mode = 2;
    p->S.rec_cap_ints = cap;
    p->S.kron = want_kron;
    if (p->S.build(p->M)) { fprintf(stderr, "qtos: %s\n", p->S.err.c_str()); delete p; return -1; }
    if (p->S.kron) {
      // the Kronecker assembly needs the benchmark's shape: a 128-slot front, no continuation records, its scratch within the LDS
      int n_cont = 0;
      for (int k = 0; k < p->S.n_stages; ++k) n_cont += p->S.srec[p->S.srec_off[k] + 6];
      const size_t need = kkt2_lds_bytes(p->S.front, p->S.n_stages, p->S.max_srec, p->S.max_drec, p->S.n_cells) + 16 + sizeof(double) * Symbolic::KRON_SM * (size_t)p->S.max_kblocks;
      if (p->S.front != 128 || n_cont != 0 || p->S.max_kblocks == 0 || need > 160 * 1024 - 256) {
        if (env.debug) fprintf(stderr, "qtos: Kronecker assembly not applicable (front %d, %d continuation records, %zu B of LDS)\n", p->S.front, n_cont, need);
        want_kron = false;
        p->M = HostModel(); p->S = Symbolic();
        p->S.env = env;
        if ((p->M.order_rule = order_rule, p->M.build(*params))) { delete p; return -1; }
        p->S.cell_mode = 2; p->S.rec_cap_ints = cap;
        if (p->S.build(p->M)) { delete p; return -1; }
      }
    }
    if (kkt2_lds_bytes(p->S.front, p->S.n_stages, p->S.max_srec, p->S.max_drec, p->S.n_cells) <= 160 * 1024 - 256) break;
  }
  const HostModel &M = p->M;
  const Symbolic &S = p->S;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) {
    fprintf(stderr, "qtos: no HIP device %d (found %d) -- the planner has no CPU fallback\n", device, ndev);
    delete p;
    return -2;
  }
  int rc = 0;
#define TRY(x) do { if ((rc = (x))) { fprintf(stderr, "qtos: %s\n", p->err.c_str()); qtos_planner_destroy(p); return rc; } } while (0)
  {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete p; return -2; }
  }
  DevPlan &D = p->dp;
  std::memset(&D, 0, sizeof(D));
  D.n_vars = M.n_vars; D.n_cons = M.n_cons; D.n_stages = S.n_stages; D.front = S.front;
  D.n_sol = M.n_sol; D.n_rec = (int)M.rec_var.size();
  {  // k_step writes a recovered node step next to the coefficient steps it is formed from: the two sets must not meet
    std::vector<char> is_rec(M.n_sol, 0);
    for (int v : M.rec_var) is_rec[v] = 1;
    for (int c : M.rec_col)
      if (c >= 0 && is_rec[c]) { p->err = "reduced base: a recovered node value is the source of another"; fprintf(stderr, "qtos: %s\n", p->err.c_str()); delete p; return -1; }
  }
  TRY(p->upload(M.rec_var, &D.rec_var)); TRY(p->upload(M.rec_col, &D.rec_col)); TRY(p->upload(M.rec_w, &D.rec_w));
  D.n_coef = M.reduce_base ? M.n_coef : 0; D.n_pz = (int)M.pz_var.size();
  TRY(p->upload(M.pc_var, &D.pc_var)); TRY(p->upload(M.pc_w, &D.pc_w));
  D.n_psw = (int)M.psw_var.size();
  TRY(p->upload(M.psw_var, &D.psw_var)); TRY(p->upload(M.psw_src, &D.psw_src)); TRY(p->upload(M.psw_w, &D.psw_w));
  TRY(p->upload(M.pz_var, &D.pz_var)); TRY(p->upload(M.pz_col, &D.pz_col)); TRY(p->upload(M.pz_w, &D.pz_w));
  D.n_dyn = (int)M.dyn.size(); D.n_rom = (int)M.rom.size(); D.n_terr = (int)M.terr.size();
  D.n_force = (int)M.force.size(); D.n_lin = (int)M.linrow.size(); D.n_blocks = (int)M.blocks.size();
  TRY(p->upload(M.dyn, &D.dyn)); TRY(p->upload(M.rom, &D.rom));
  {  // flat tables of the spline inputs in pre-pass order (the VecIn records lead the instance structs)
    auto flat = [&](const void *inst0, size_t inst_bytes, size_t n_inst, int n_vec, const i4_t **var, const d2_t **wa, const d2_t **wb) {
      std::vector<i4_t> v;
      std::vector<d2_t> a, b;
      for (size_t i = 0; i < n_inst; ++i)
        for (int k = 0; k < n_vec; ++k) {
          const VecIn &in = ((const VecIn *)((const char *)inst0 + i * inst_bytes))[k];
          for (int d = 0; d < 3; ++d) {
            v.push_back(i4_t{in.var[d], in.var[3 + d], in.var[6 + d], in.var[9 + d]});
            a.push_back(d2_t{in.w[0], in.w[1]});
            b.push_back(d2_t{in.w[2], in.w[3]});
          }
        }
      int rc2 = p->upload(v, var);
      if (!rc2) rc2 = p->upload(a, wa);
      if (!rc2) rc2 = p->upload(b, wb);
      return rc2;
    };
    static_assert(DYN_VIN == 39 && ROM_VIN == 9, "13 / 3 input vectors of three components");
    TRY(flat(M.dyn.data(), sizeof(DynInst), M.dyn.size(), 13, &D.pre_dyn_var, &D.pre_dyn_wa, &D.pre_dyn_wb));
    TRY(flat(M.rom.data(), sizeof(RomInst), M.rom.size(), 3, &D.pre_rom_var, &D.pre_rom_wa, &D.pre_rom_wb));
  }
  TRY(p->upload(M.force, &D.force)); TRY(p->upload(M.linrow, &D.lin));
  {  // entry lists of the iterate-dependent Jacobian values, packed (model.hpp: PackedTerm1 / PackedTerm3)
    std::map<unsigned long long, int> idx_of;
    std::vector<double> coefs;
    bool fits = true;
    auto ci = [&](double a) {
      unsigned long long bits;
      std::memcpy(&bits, &a, 8);
      auto it = idx_of.find(bits);
      if (it == idx_of.end()) { it = idx_of.emplace(bits, (int)coefs.size()).first; coefs.push_back(a); }
      return (unsigned short)it->second;
    };
    auto off16 = [&](int off) { if (off < 0 || off > 65535) fits = false; return (unsigned short)off; };
    std::vector<PackedTerm1> d1, r1;
    std::vector<PackedTerm3> d3;
    for (const LinTerm1 &t : S.dyn_t1) d1.push_back({(unsigned)t.pos, off16(t.off), ci(t.a)});
    for (const LinTerm1 &t : S.rom_t1) r1.push_back({(unsigned)t.pos, off16(t.off), ci(t.a)});
    for (const LinTerm3 &t : S.dyn_t3)
      d3.push_back({(unsigned)t.pos, {off16(t.off[0]), off16(t.off[1]), off16(t.off[2])}, {ci(t.a[0]), ci(t.a[1]), ci(t.a[2])}});
    if (!fits || coefs.size() > 65535) {
      p->err = "Jacobian entry lists exceed the packed format (16-bit offsets / coefficient indices)";
      fprintf(stderr, "qtos: %s\n", p->err.c_str());
      qtos_planner_destroy(p);
      return -4;
    }
    D.n_lin_coef = (int)coefs.size();
    TRY(p->upload(d1, &D.dyn_t1)); TRY(p->upload(d3, &D.dyn_t3)); TRY(p->upload(r1, &D.rom_t1)); TRY(p->upload(coefs, &D.lin_coef));
  }
  TRY(p->upload(S.dyn_t1_off, &D.dyn_t1_off)); TRY(p->upload(S.dyn_t3_off, &D.dyn_t3_off));
  D.n_rom_t1 = (int)S.rom_t1.size();
  D.dyn_chunk = M.dyn_chunk;
  D.rom_chunk = S.rom_chunk;
  TRY(p->upload(S.rom_t1_off, &D.rom_t1_off));
  TRY(p->upload(S.amask, &D.amask));
  TRY(p->upload(S.amask2, &D.amask2));
  TRY(p->upload(S.nxt_pack, &D.nxt_pack));
  TRY(p->upload(S.ctab, &D.ctab));
  TRY(p->upload(S.rtab, &D.rtab));
  TRY(p->upload(S.rhs_ptr, &D.rhs_ptr)); TRY(p->upload(S.rhs_gpos, &D.rhs_gpos)); TRY(p->upload(S.rhs_row, &D.rhs_row));
  TRY(p->upload(S.kx_ptr, &D.kx_ptr)); TRY(p->upload(S.kx_col, &D.kx_col)); TRY(p->upload(S.kx_pos, &D.kx_pos));
  D.n_unknowns = S.n_unknowns;
  D.chord_tol = M.P.chord_tol;
  D.chord_max = M.P.chord_max > 0 ? M.P.chord_max : 1;
  D.chord_shrink = M.P.chord_shrink > 0 ? M.P.chord_shrink : 1.0 / 3.0;
  D.n_cells = S.n_cells;
  D.n_cont = 0;
  for (int k = 0; k < S.n_records; ++k) D.n_cont += S.srec[S.srec_off[k] + 6];
  D.table = nullptr; D.tab_dx = D.tab_dy = nullptr; D.tab_ndx = D.tab_ndy = 0;
  D.off_lin = M.off_lin; D.off_ang = M.off_ang;
  for (int e = 0; e < NEE; ++e) D.off_eem[e] = M.off_eem[e];
  TRY(p->upload(S.cont, &D.cont));
  {  // terrain rows with their stream positions: the row's Jacobian entries, and the pivot diagonals of the foot node's x and
     // y (two-phase solve: stance rows only)
    std::vector<int> dpos_of_var(M.n_sol, -1);
    for (int i = 0; i < (int)S.pack_src.size(); ++i)
      if ((S.pack_src[i] >> 28) == 5) {
        const int u = S.piv_unknown[S.pack_src[i] & 0x0fffffff];
        if (u >= 0 && u < M.n_sol) dpos_of_var[u] = i;
      }
    std::vector<TerrDev> td;
    for (const TerrInst &t : M.terr) {
      TerrDev d = {t.vx, t.vy, t.vz, t.row, -1, -1, -1, -1, -1, 0, 0.0, 0.0};
      if (t.in_kkt) {
        const bool stance = M.row_kind[t.row] == 1;   // equality block: entries at their own stream positions
        auto pos = [&](int c) { return c < 0 ? -1 : (stance ? S.eq_pos[t.goff + c] : t.goff + c); };
        d.px = pos(t.cx); d.py = pos(t.cy); d.pz = pos(t.cz);
        if (M.P.hold_from > 0 && stance) {
          d.d0 = dpos_of_var[t.vx]; d.d1 = dpos_of_var[t.vy];
          d.ex = M.sol_diag[t.vx] - M.P.delta_x; d.ey = M.sol_diag[t.vy] - M.P.delta_x;
        }
      }
      td.push_back(d);
    }
    TRY(p->upload(td, &D.terr));
    D.hold_from = M.P.hold_from;
    D.spec_jac = env.spec_jac;
    D.hold_weight = M.P.hold_weight > 0 ? M.P.hold_weight : 1e6;
    D.hold_tol = M.P.hold_tol;
  }
  TRY(p->upload(M.blocks, &D.blocks)); TRY(p->upload(M.block_cols, &D.block_cols));
  {
    std::vector<IqRow> rows;   // blocks carry their stream offsets once the symbolic analysis has run
    for (const Block &b : M.blocks)
      if (b.kind == 1)
        for (int r = 0; r < b.m; ++r) rows.push_back({b.goff + r * b.n, b.n, b.col_off, b.row0 + r});
    D.n_iq_rows = (int)rows.size();
    TRY(p->upload(rows, &D.iq_rows));
  }
  {
    std::vector<SwTask> tasks;
    std::vector<int> cpos, c16;
    D.sw_steps = build_sweep_tasks(M, S, tasks, cpos, c16);
    const bool ok = D.sw_steps > 0;
    D.sw_on = ok && D.n_iq_rows > 0 && S.front / PIV < SW_W0 && env.sweep_ds != 0;   // (waves 13 .. 15 must be without rows)
    TRY(p->upload(tasks, &D.sw_tasks)); TRY(p->upload(cpos, &D.sw_cpos)); TRY(p->upload(c16, &D.sw_c16));
  }
  {
    std::vector<int> iq, eq;
    std::vector<double> lo, hi;
    for (int r = 0; r < M.n_cons; ++r) {
      if (M.row_kind[r] == 2) { iq.push_back(r); lo.push_back(M.con_lo[r]); hi.push_back(M.con_hi[r]); }
      else if (M.row_kind[r] == 1) eq.push_back(r);
    }
    D.n_iq = (int)iq.size(); D.n_eqw = (int)eq.size();
    TRY(p->upload(iq, &D.iq_idx)); TRY(p->upload(eq, &D.eq_idx));
    TRY(p->upload(lo, &D.iq_lo)); TRY(p->upload(hi, &D.iq_hi));
  }
  TRY(p->upload(M.g_static, &D.g_static));
  TRY(p->upload(S.piv_slot, &D.piv_slot)); TRY(p->upload(S.piv_unknown, &D.piv_unknown));
  TRY(p->upload(S.piv_diag, &D.piv_diag)); TRY(p->upload(S.stages, &D.stages));
  TRY(p->upload(S.eq_entries, &D.eq_entries)); TRY(p->upload(S.eq_rhs, &D.eq_rhs));
  TRY(p->upload(S.iq_blocks, &D.iq_blocks)); TRY(p->upload(S.iq_slots, &D.iq_slots));
  TRY(p->upload(M.con_lo, &D.con_lo)); TRY(p->upload(M.con_hi, &D.con_hi));
  TRY(p->upload(M.row_kind, &D.row_kind)); TRY(p->upload(M.init, &D.init));
  TRY(p->upload(M.node_time, &D.var_time));   // (node times, not the order's keys: model.hpp node_time)
  D.max_stage_g = S.max_stage_g;
  TRY(p->upload(S.srec, &D.srec)); TRY(p->upload(S.srec_off, &D.srec_off));
  TRY(p->upload(S.pack_src, &D.pack_src)); TRY(p->upload(S.drec_off, &D.drec_off));
  TRY(p->upload(S.diag_pos, &D.diag_pos));
  TRY(p->upload(S.pair_groups, &D.pair_groups));
  TRY(p->upload(S.eq_pos, &D.eq_pos)); TRY(p->upload(S.rhs_pos, &D.rhs_pos));
  TRY(p->upload(S.sig_pos, &D.sig_pos)); TRY(p->upload(S.w_pos, &D.w_pos));
  D.max_srec = S.max_srec; D.max_drec = S.max_drec; D.stream_len = (int)S.pack_src.size();
  D.mass = M.P.mass; D.gravity = M.P.gravity; D.mu_fric = M.P.mu; D.f_max = M.P.f_max; D.T = M.T;
  for (int i = 0; i < 9; ++i) D.Ib[i] = M.P.inertia_b[i];
  for (int e = 0; e < NEE; ++e)
    for (int d = 0; d < 3; ++d) D.nominal[e][d] = M.P.nominal_stance[e][d];
  D.tol = M.P.tol; D.mu_init = M.P.mu_init; D.mu_min = M.P.mu_min; D.delta_x = M.P.delta_x;
  D.mu_superlinear = M.P.mu_superlinear != 0;
  D.eps_dual = M.P.eps_dual; D.max_iter = M.P.max_iter; D.stall_iters = M.P.stall_iters; D.stall_alpha = M.P.stall_alpha;
  D.slack_push = M.P.slack_push > 0 ? M.P.slack_push : 0.01;
  D.warm_slack_push = M.P.warm_slack_push > 0 ? M.P.warm_slack_push : 0.01;
  D.terrain_mode = M.P.terrain_mode;
  D.g_doubles = S.g_doubles;
  D.panel_stride = (long long)S.n_stages * (S.front + 1) * PIV;
  // LDS budget of k_kkt
  const int F = S.front;
  p->kkt_lds = p->use_kkt5 ? kkt5_lds_bytes(F, S.n_stages, S.max_srec, S.max_drec, S.n_cells) : p->use_kkt3 ? kkt3_lds_bytes(F, S.n_stages, S.max_srec, S.max_drec, S.n_cells) : kkt2_lds_bytes(F, S.n_stages, S.max_srec, S.max_drec, S.n_cells);
  D.kron_lds_off = 0;
  if (env.debug) fprintf(stderr, "qtos: Kronecker assembly %s (kkt3 %d, most blocks in a record %d)\n", S.kron ? "on" : "off", (int)p->use_kkt3, S.max_kblocks);
  if (S.kron && !p->use_kkt3) {
    D.kron_lds_off = (int)((p->kkt_lds + 15) & ~(size_t)15);
    p->kkt_lds = (size_t)D.kron_lds_off + sizeof(double) * Symbolic::KRON_SM * (size_t)S.max_kblocks;
  }
  if (p->use_kkt5 && F > 128) D.sw_on = 0;   // (nine tile waves of twelve: fewer than three helper waves are left)
  if (D.sw_on) {
    // the helper waves' tables of the backward sweep (solution by position, rounds) behind the sweep's own: within the LDS
    // the forward pass needs anyway, or the kernel's allocation grows up to the limit; beyond that k_step forms ds itself
    const size_t need = (p->use_kkt5 ? kkt5_sweep_base_bytes(F, S.n_stages) : kkt2_sweep_base_bytes(F, S.n_stages)) + sweep_ds_lds_bytes(S.n_stages, D.sw_steps);
    if (need > 160 * 1024 - 256 || chord_lds_bytes(S.n_stages, D.sw_steps) > 96 * 1024) D.sw_on = 0;
    else p->kkt_lds = std::max(p->kkt_lds, need);
  }
  p->kkt_threads = p->use_kkt5 ? KT5 : KT2;
  const int max_front = 208;
  if (!p->use_kkt5 && (S.max_drec > 2 * 2 * KT || S.max_srec > 3 * 4 * KT || F > max_front || (S.pack_src.size() & 1))) {
    p->err = "stage record exceeds the prefetch registers";
    fprintf(stderr, "qtos: stage records too long (%d doubles, %d ints) or front %d > %d\n", S.max_drec, S.max_srec, F, max_front);
    qtos_planner_destroy(p);
    return -4;
  }
  if (env.debug) fprintf(stderr, "qtos: front %d, k_kkt LDS %zu B\n", F, p->kkt_lds);
  if (p->kkt_lds > 160 * 1024 - 256) {
    p->err = "front too large for LDS";
    fprintf(stderr, "qtos: front %d needs %zu B of LDS\n", F, p->kkt_lds);
    qtos_planner_destroy(p);
    return -4;
  }
  {
    p->kkt_fn = p->use_kkt5 ? kkt5_kernel(F) : p->use_kkt3 ? kkt3_kernel(F) : kkt2_kernel(F, D.n_cont > 0, S.kron);
    p->chord_fn = chord_kernel(F);
    if (!p->kkt_fn) { p->err = "no k_kkt instantiation for this front size"; qtos_planner_destroy(p); return -4; }
    hipError_t e = hipFuncSetAttribute((const void *)p->kkt_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->kkt_lds);
    if (e != hipSuccess) { p->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); fprintf(stderr, "qtos: %s\n", p->err.c_str()); qtos_planner_destroy(p); return -2; }
    if (p->chord_fn && D.sw_on) {
      e = hipFuncSetAttribute((const void *)p->chord_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chord_lds_bytes(S.n_stages, D.sw_steps));
      if (e != hipSuccess) { p->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); fprintf(stderr, "qtos: %s\n", p->err.c_str()); qtos_planner_destroy(p); return -2; }
    }
  }
  p->eval_lds = sizeof(double) * (((size_t)M.n_sol + 1) / 2 * 2 + std::max((size_t)DYN_LOC * D.dyn_chunk, (size_t)ROM_LOC * D.rom_chunk) +
                                  std::max((size_t)DYN_VIN * D.dyn_chunk, (size_t)ROM_VIN * D.rom_chunk));
  // the coefficient table of the entry lists joins the scratch when it fits (schedules with irregular phase durations have
  // ten thousand distinct Hermite weights: those read it from memory)
  D.coef_in_lds = p->eval_lds + sizeof(double) * (size_t)D.n_lin_coef <= 150 * 1024;
  if (D.coef_in_lds) p->eval_lds += sizeof(double) * (size_t)D.n_lin_coef;
  {  // k_step's pass over the chord right-hand side: unknown sums and list bounds, then the factors of rhs_chunk entries
    D.n_rhs_ent = (int)S.rhs_gpos.size();
    const size_t nuk = (size_t)S.n_unknowns, fixed = ((nuk + 1) & ~(size_t)1) + ((nuk + 4) & ~(size_t)3) / 2;
    const size_t room = 150 * 1024 / sizeof(double) > fixed ? (150 * 1024 / sizeof(double) - fixed) / (2 * (size_t)ET) : 0;   // passes of ET entries that fit
    D.rhs_chunk = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((D.n_rhs_ent + ET - 1) / ET, 12), room)) * ET;
    p->eval_lds = std::max(p->eval_lds, (fixed + 2 * (size_t)D.rhs_chunk) * sizeof(double));
  }
  if (p->eval_lds > 150 * 1024) {
    p->err = "too many dynamics knots for the LDS scratch";
    fprintf(stderr, "qtos: evaluation kernels need %zu B of LDS\n", p->eval_lds);
    qtos_planner_destroy(p);
    return -4;
  }
  for (const void *fn : {(const void *)k_start, (const void *)k_step, (const void *)k_debug_eval})
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->eval_lds) != hipSuccess) { qtos_planner_destroy(p); return -2; }
  // sampling tables
  SamplePlan &SP = p->sp;
  std::memset(&SP, 0, sizeof(SP));
  SP.n_vars = M.n_vars; SP.T = M.T;
  TRY(upload_spline(p, M.lin, &SP.lin)); TRY(upload_spline(p, M.ang, &SP.ang));
  for (int e = 0; e < NEE; ++e) { TRY(upload_spline(p, M.eem[e], &SP.eem[e])); TRY(upload_spline(p, M.eef[e], &SP.eef[e])); }
  // workspaces
  DevWork &W = p->wk;
  std::memset(&W, 0, sizeof(W));
  const size_t Bm = (size_t)max_batch, n = M.n_vars, m = M.n_cons;
  TRY(p->alloc(&W.x, Bm * n)); TRY(p->alloc(&W.dx, Bm * (size_t)M.n_sol));
  TRY(p->alloc(&W.g, Bm * m)); TRY(p->alloc(&W.gt, Bm * m)); TRY(p->alloc(&W.s, Bm * m));
  TRY(p->alloc(&W.zl, Bm * m)); TRY(p->alloc(&W.zu, Bm * m)); TRY(p->alloc(&W.ds, Bm * m));
  TRY(p->alloc(&W.dzl, Bm * m)); TRY(p->alloc(&W.dzu, Bm * m)); TRY(p->alloc(&W.sig, Bm * m));
  TRY(p->alloc(&W.w, Bm * m));
  TRY(p->alloc(&W.panel, Bm * (size_t)D.panel_stride));
  TRY(p->alloc(&W.stream, Bm * (size_t)S.pack_src.size() + 64));   // (+ 64: the helper waves of the sweep read whole groups of a row's entries, sw_load)
  {  // constants of the stream (static Jacobian values, pivot diagonals): written once per problem
    std::vector<double> one(S.pack_src.size(), 0.0);
    for (size_t i = 0; i < S.const_pos.size(); ++i) one[S.const_pos[i]] = S.const_val[i];
    for (size_t b = 0; b < Bm; ++b)
      HIPCHK(p, hipMemcpy(W.stream + b * one.size(), one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  TRY(p->alloc(&W.mu, Bm)); TRY(p->alloc(&W.viol, Bm));
  TRY(p->alloc(&W.best_viol, Bm)); TRY(p->alloc(&W.best_it, Bm)); TRY(p->alloc(&W.xbest, Bm * n));
  TRY(p->alloc(&W.held, Bm));
#ifdef QTOS_STAMPS
  if (M.P.max_iter < 79) { p->err = "diagnostic build (QTOS_STAMPS) parks its stamps in trace rows 16..79: create the planner with max_iter >= 79"; fprintf(stderr, "qtos: %s\n", p->err.c_str()); qtos_planner_destroy(p); return -1; }
#endif
  TRY(p->alloc(&W.trace, Bm * (size_t)(M.P.max_iter + 1) * 4));
  TRY(p->alloc(&W.status, Bm)); TRY(p->alloc(&W.iters, Bm)); TRY(p->alloc(&W.done, Bm));
  TRY(p->alloc(&W.n_active, 4 * 4));   // four counters per lane
  TRY(p->alloc(&W.chord, Bm));
  TRY(p->alloc(&W.chord_run, Bm));
  TRY(p->alloc(&W.jam, Bm));
  TRY(p->alloc(&W.rhs, Bm * (size_t)S.n_unknowns));
  TRY(p->alloc(&W.minv, Bm * (size_t)S.n_stages * PIV * PIV));
  TRY(p->alloc(&W.sol, Bm * (size_t)S.n_stages * PIV)); TRY(p->alloc(&W.sol0, Bm * (size_t)S.n_stages * PIV));
  TRY(p->alloc(&W.dx0, Bm * (size_t)M.n_sol)); TRY(p->alloc(&W.ur, Bm * m));
  TRY(p->alloc(&p->d_start, Bm * QTOS_START_DOUBLES)); TRY(p->alloc(&p->d_goal, Bm * 3));
  TRY(p->alloc(&p->d_nodes, Bm * n)); TRY(p->alloc(&p->d_warm, Bm * n)); TRY(p->alloc(&p->d_map, Bm));
  if (hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) { qtos_planner_destroy(p); return -2; }
  if (hipMalloc(&p->d_totals, 2 * sizeof(long long)) != hipSuccess || hipMemset(p->d_totals, 0, 2 * sizeof(long long)) != hipSuccess) { qtos_planner_destroy(p); return -3; }
  p->last_stream = p->own_stream;
  // launch slots of a call: its iterations plus the launches some problem sat out (at most one per blind slot)
  p->max_slots = 2 * M.P.max_iter + 2;
  {  // lanes: one per part of a call that is larger than the GPU -- QTOS_LANES: at most that many; default ONE: measured at 1024
     // problems per call (round 4, profiles/r04_lanes.txt), parts that start together also reach their stragglers together, and
     // four lock-step loops of 256 pay four tails where one loop of 1024 pays one (exp_5 75.8 K plans/s on four lanes, 84.0 K
     // on two, 84.2 K on one)
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    p->lane_chunk = cus;
    const int want = env.lanes;
    const int n_lanes = std::max(1, std::min(want, (max_batch + cus - 1) / cus));
    p->lanes.resize(n_lanes);
    if (hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) != hipSuccess) { qtos_planner_destroy(p); return -2; }
    for (int j = 0; j < n_lanes; ++j) {
      QtosPlanner::Lane &L = p->lanes[j];
      if (hipHostMalloc((void **)&L.h_active, 4 * sizeof(int) * ((size_t)p->max_slots + 1), hipHostMallocMapped) != hipSuccess) { qtos_planner_destroy(p); return -3; }
      if (hipHostGetDevicePointer((void **)&L.h_active_dev, L.h_active, 0) != hipSuccess) { qtos_planner_destroy(p); return -2; }
      L.was_kkt.assign(p->max_slots + 1, 0);
      L.was_chord.assign(p->max_slots + 1, 0);
      L.d_n_active = W.n_active + 4 * j;
      if (j > 0 && hipStreamCreateWithFlags(&L.own, hipStreamNonBlocking) != hipSuccess) { qtos_planner_destroy(p); return -2; }
      if (hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking) != hipSuccess) { qtos_planner_destroy(p); return -2; }
      if (hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming) != hipSuccess) { qtos_planner_destroy(p); return -2; }
      L.ev.resize(5 * (size_t)p->max_slots + 3);
      for (auto &e : L.ev)
        if (hipEventCreate(&e) != hipSuccess) { qtos_planner_destroy(p); return -2; }
    }
  }
#undef TRY
  *out = p;
  return 0;
}

static void fill_dims(const HostModel &M, const Symbolic &S, QtosDims *d) {
  std::memset(d, 0, sizeof(*d));
  d->n_vars = M.n_vars; d->n_cons = M.n_cons;
  d->n_free = 0;
  for (int v = 0; v < M.n_vars; ++v) d->n_free += M.is_free(v) ? 1 : 0;   // (the NLP's count; with reduce_base the KKT system has fewer: n_unknowns)
  for (int r = 0; r < M.n_cons; ++r) {
    const bool eq = M.con_lo[r] == M.con_hi[r];
    if (eq) d->n_eq++;
    else {
      d->n_ineq++;
      const bool hl = M.con_lo[r] > -1e19, hu = M.con_hi[r] < 1e19;
      if (hl && hu) d->n_ineq_both++;
      else if (hl) d->n_ineq_lower++;
      else d->n_ineq_upper++;
    }
  }
  d->n_eq_work = S.n_eq; d->n_unknowns = S.n_real_unknowns; d->n_stages = S.n_stages;   // (the unknowns of the KKT system; positions incl. the dummy pivots of short stages: n_stages x 16)
  d->pivots = PIV; d->front = S.front;
  d->n_base_nodes = M.n_base_nodes; d->n_dyn_times = (int)M.t_dyn.size(); d->n_rom_times = (int)M.t_rom.size();
  d->n_rows_csv = (int)std::llround(M.T * 1000.0) + 1;
  d->panel_doubles = (long long)S.n_stages * (S.front + 1) * PIV; d->g_doubles = S.g_doubles;
  d->kkt_algorithmic_bytes = S.algorithmic_bytes; d->kkt_flops = S.flops;
  d->envelope = S.envelope; d->max_active = S.max_active; d->order_rule = M.order_rule;
  d->duration = M.T;
}

int qtos_planner_dims(const QtosPlanner *p, QtosDims *d) {
  if (!p || !d) return -1;
  fill_dims(p->M, p->S, d);
  return 0;
}

int qtos_analyze(const QtosParams *params, QtosDims *d, int *stage_active, int max_stages) {
  if (!params || !d) return -1;
  HostModel M;
  Symbolic S;
  S.env = QtosEnv::parse();
  if ((M.order_rule = pick_order_rule(*params, S.env), M.build(*params))) { fprintf(stderr, "qtos: %s\n", M.err.c_str()); return -1; }
  if (S.env.debug_kron) S.kron = true;
  if (S.build(M)) { fprintf(stderr, "qtos: %s\n", S.err.c_str()); return -1; }
  if (S.env.debug_kron) {
    int nb = 0;
    const double worst = S.check_kron(&nb);
    size_t tot = 0;
    for (const Block &b : M.blocks) tot += b.kind == 1;
    fprintf(stderr, "qtos: Kronecker blocks %d of %zu inequality blocks, most in a record %d, worst relative difference %.2e, max record %d ints / %d doubles\n", nb, tot, S.max_kblocks, worst, S.max_srec, S.max_drec);
  }
  fill_dims(M, S, d);
  if (S.env.debug) { std::vector<SwTask> t; std::vector<int> c, c2; (void)build_sweep_tasks(M, S, t, c, c2); }
  if (stage_active)
    for (int k = 0; k < S.n_stages && k < max_stages; ++k) stage_active[k] = S.stages[k].n_active;
  return 0;
}

// Host-only: the elimination order by position (qtos_debug_structure's `order` without a planner handle): var index, n_vars + row
// for a multiplier, -1 for a dummy pivot; returns the number of positions (n_stages * pivots) or < 0.
int qtos_analyze_order(const QtosParams *params, int *order, int max_positions) {
  if (!params || !order) return -1;
  HostModel M;
  Symbolic S;
  S.env = QtosEnv::parse();
  if ((M.order_rule = pick_order_rule(*params, S.env), M.build(*params))) { fprintf(stderr, "qtos: %s\n", M.err.c_str()); return -1; }
  if (S.build(M)) { fprintf(stderr, "qtos: %s\n", S.err.c_str()); return -1; }
  const int np = S.n_stages * PIV;
  for (int i = 0; i < np && i < max_positions; ++i) order[i] = i < (int)S.order.size() ? S.order[i] : -1;   // (as qtos_debug_structure)
  return np;
}

// Host-only: what a TWO-ENDED elimination of this model's KKT matrix would look like (Symbolic::analyze_two_ended: a chain from
// t = 0 forward, a chain from t = T backward, the unknowns alive across the split time last), and the LDS a workgroup that runs
// both chains would need with the layouts of today's kernel.  out (>= 20 ints):
//   [0] stages today  [1] front today  [2] split stage  [3] stages of chain L  [4] of chain R  [5] separator unknowns
//   [6] separator stages  [7] front of L  [8] of R  [9] of the separator  [10] serial steps  [11] populated peak of L  [12] of R
//   LDS bytes: [13] today's kernel for this model, of which [14] panels, [15] record buffers, [16] cells;
//   [17] two chains with today's layouts (panels, cells and both record buffers twice, the fixed part twice)
//   [18] two chains, LEAN: per chain two panels + the blanked copy, one record buffer holding the dynamic record only (gather
//        tables read from L2), the cells; the fixed part twice   [19] the limit (160 KB - 256 B)
int qtos_analyze_two_ended(const QtosParams *params, int *out, int n_out) {
  if (!params || !out || n_out < 20) return -1;
  HostModel M;
  Symbolic S;
  S.env = QtosEnv::parse();
  if ((M.order_rule = pick_order_rule(*params, S.env), M.build(*params))) { fprintf(stderr, "qtos: %s\n", M.err.c_str()); return -1; }
  if (S.build(M)) { fprintf(stderr, "qtos: %s\n", S.err.c_str()); return -1; }
  const Symbolic::TwoEnded t = S.analyze_two_ended(M);
  const int F = S.front, Fc = std::max(t.front_left, t.front_right);
  auto panel_b = [](int f, int n_pan) { return (size_t)((f + 1) * PLD * n_pan + f * PLD) * sizeof(double); };
  const size_t rec_b = 2 * (kkt2_dbuf_doubles(F, S.max_drec) * sizeof(double) + kkt2_sbuf_ints(F, S.max_srec) * sizeof(int));
  const size_t cells_b = (((size_t)S.n_cells + 1) & ~(size_t)1) * sizeof(double);
  const size_t total = kkt2_lds_bytes(F, S.n_stages, S.max_srec, S.max_drec, S.n_cells);
  const size_t fixed_b = total > panel_b(F, 3) + rec_b + cells_b ? total - panel_b(F, 3) - rec_b - cells_b : 0;
  out[0] = S.n_stages; out[1] = F; out[2] = t.split_stage; out[3] = t.stages_left; out[4] = t.stages_right; out[5] = t.sep_unknowns;
  out[6] = t.stages_sep; out[7] = t.front_left; out[8] = t.front_right; out[9] = t.front_sep; out[10] = t.serial_steps;
  out[11] = t.peak_left; out[12] = t.peak_right;
  out[13] = (int)total; out[14] = (int)panel_b(F, 3); out[15] = (int)rec_b; out[16] = (int)cells_b;
  out[17] = (int)(2 * (panel_b(Fc, 3) + rec_b + cells_b + fixed_b));
  out[18] = (int)(2 * (panel_b(Fc, 2) + kkt2_dbuf_doubles(F, S.max_drec) * sizeof(double) + cells_b + fixed_b));
  out[19] = 160 * 1024 - 256;
  return 0;
}

// Host-only: the Kronecker structure of the range-of-motion blocks (Symbolic::kron_meta, QTOS_KRON): how many inequality
// blocks have it, the most in one stage record, and the largest relative difference between an entry of G' S G / G' w formed
// through the 33 sums of a block and the direct three-term sum, on random matrices and weights (Symbolic::check_kron).
int qtos_analyze_kron(const QtosParams *params, int *n_blocks, int *n_kron, int *max_in_record, double *worst) {
  if (!params || !n_blocks || !n_kron || !max_in_record || !worst) return -1;
  HostModel M;
  Symbolic S;
  S.env = QtosEnv::parse();
  if ((M.order_rule = pick_order_rule(*params, S.env), M.build(*params))) { fprintf(stderr, "qtos: %s\n", M.err.c_str()); return -1; }
  S.kron = true;
  if (S.build(M)) { fprintf(stderr, "qtos: %s\n", S.err.c_str()); return -1; }
  *n_blocks = 0;
  for (const Block &b : M.blocks) *n_blocks += b.kind == 1;
  *worst = S.check_kron(n_kron);
  *max_in_record = S.max_kblocks;
  return 0;
}

// Host-only: the schedule the helper waves of the backward sweep follow (build_sweep_tasks).  Per place of a round (16 per
// round, round i runs in step i of the chain, i.e. while it solves stage n_stages - 1 - i): the constraint row (-1: empty),
// the row's entries, and the smallest and largest position (elimination order) of the row's columns.
int qtos_analyze_sweep(const QtosParams *params, int *n_rounds, int *rows, int *entries, int *pos_min, int *pos_max, int max_places) {
  if (!params || !n_rounds) return -1;
  HostModel M;
  Symbolic S;
  S.env = QtosEnv::parse();
  if ((M.order_rule = pick_order_rule(*params, S.env), M.build(*params))) { fprintf(stderr, "qtos: %s\n", M.err.c_str()); return -1; }
  if (S.build(M)) { fprintf(stderr, "qtos: %s\n", S.err.c_str()); return -1; }
  std::vector<SwTask> tasks;
  std::vector<int> cpos, c16;
  *n_rounds = build_sweep_tasks(M, S, tasks, cpos, c16);
  if (*n_rounds <= 0) return -4;
  for (int i = 0; i < *n_rounds * SW_ROUND && i < max_places; ++i) {
    const SwTask &t = tasks[i];
    int lo = INT_MAX, hi = -1;
    for (int a = 0; a < t.n && t.row >= 0; ++a) { lo = std::min(lo, cpos[t.cpos_off + a]); hi = std::max(hi, cpos[t.cpos_off + a]); }
    if (rows) rows[i] = t.row;
    if (entries) entries[i] = t.row >= 0 ? t.n : 0;
    if (pos_min) pos_min[i] = t.row >= 0 ? lo : -1;
    if (pos_max) pos_max[i] = hi;
    // the 16-bit copy of the positions the lanes read must say the same as the list
    if (t.row >= 0)
      for (int q = 0; q < 4; ++q)
        for (int u = 0; u < SW_RU; ++u) {
          const int n4 = t.n & ~3, e = std::max(std::min(q + 4 * u, n4 - 4 + q), 0);
          const unsigned w = (unsigned)c16[t.c16_off + 4 * q + (u >> 1)];
          if ((int)((w >> (16 * (u & 1))) & 0xffffu) != cpos[t.cpos_off + std::min(e, t.n - 1)]) return -5;
        }
  }
  return 0;
}

int qtos_set_heightfields(QtosPlanner *p, int n_maps, const double *height, int hnx, int hny,
                          double cell, double x0, double y0) {
  if (!p) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  if (p->d_height) {
    (void)hipFree(p->d_height);
    p->d_height = nullptr;
  }
  p->dp.height = nullptr;
  p->dp.n_maps = 0;
  if (n_maps <= 0 || !height) return 0;
  if (hnx < 1 || hny < 1 || !(cell > 0)) return -1;
  const size_t cnt = (size_t)n_maps * hnx * hny;
  HIPCHK(p, hipMalloc((void **)&p->d_height, cnt * sizeof(double)));
  HIPCHK(p, hipMemcpy(p->d_height, height, cnt * sizeof(double), hipMemcpyHostToDevice));
  p->dp.height = p->d_height;
  p->dp.n_maps = n_maps; p->dp.hnx = hnx; p->dp.hny = hny;
  p->dp.hcell = cell; p->dp.hx0 = x0; p->dp.hy0 = y0;
  return 0;
}

// results of a batch to the caller's buffers (one launch instead of four copies) and the running totals of
// qtos_plan_totals
// The counts of unfinished problems to the host: a store into mapped pinned memory by a one-wave kernel.  (A copy
// (hipMemcpyAsync) between two kernels of a stream makes the compute queue wait on the copy engine's signal; with more
// streams than hardware queues that wait holds up the other streams of the queue too: four sets of receding windows
// then ran one after the other.)
// The word carries the sequence number of the call that queued the launch: blind iterations a call queued beyond its end
// still store into their slots AFTER the next call of the handle has reset them (qtos_plan_submit), and a word of another
// call must read as "no counts yet", not as "nothing left to do" (24 bits per count: qtos_planner_create limits max_batch).
__global__ void k_post_counts(const int *n_active, int *host_slot, unsigned seq) {
  // the two diagnostics first (earliest launch slot a problem sat out, launch slots with work), then both counts in ONE
  // eight-byte store: the host spins on that word (qtos_plan_poll) and must never see half of it
  if (threadIdx.x == 0) {
    const unsigned long long d = (unsigned long long)(unsigned)n_active[2] | ((unsigned long long)(unsigned)n_active[3] << 32);
    *(volatile unsigned long long *)(host_slot + 2) = d;
    __threadfence_system();
    const unsigned long long v = (unsigned long long)((unsigned)n_active[0] & 0xffffffu) | ((unsigned long long)((unsigned)n_active[1] & 0xffffffu) << 24) |
                                 ((unsigned long long)(seq & 0xffffu) << 48);
    *(volatile unsigned long long *)host_slot = v;
  }
  __threadfence_system();
}

// One launch slot of the call in flight, queued on its stream: the solve kernels named by `kinds` (bit 0 the factorising
// kernel, bit 1 k_chord), then k_step, then (post) the counts to the host.
//   informed  the counts of unfinished problems (n) and of problems flagged for a chord step (nc) in front of the slot are
//             known: kinds = the kernels with work; next to a factorisation of other problems the chord solve runs on the
//             side stream (the workgroups of the factorising kernel that belong to its problems leave at once and k_chord
//             gets their CUs: the batch pays max(k_kkt, k_chord), not the sum).
//   pattern   kinds = what this slot of the handle's last calls needed (QtosPlanner::pat), queued before any count of this
//             call has come back; a problem that waits for the other kernel sits the launch out (k_step).
//   blind     both kernels (qtos_set_speculation; each leaves at once for the problems that are not its own).
static int queue_iteration(QtosPlanner *p, QtosPlanner::Lane &c, int it, int kinds, bool post) {
  const DevPlan &D = p->dp;
  hipStream_t st = c.st;
  if (!p->chord_fn) kinds &= 1;
  const bool do_kkt = kinds & 1, do_chord = (kinds & 2) != 0;
  c.was_kkt[it] = do_kkt;
  c.was_chord[it] = do_chord;
  const bool fork = do_kkt && do_chord;
  hipStream_t cs = fork ? c.side : st;
  if (fork) {
    HIPCHK(p, hipEventRecord(c.ev_fork, st));            // everything up to the previous k_step
    HIPCHK(p, hipStreamWaitEvent(cs, c.ev_fork, 0));
  }
  if (do_kkt) {
    if (p->per_kernel_events) HIPCHK(p, hipEventRecord(c.ev[2 + 5 * it], st));
    hipLaunchKernelGGL(p->kkt_fn, dim3(c.B), dim3(p->kkt_threads), p->kkt_lds, st, D, c.W, c.B);
    if (p->per_kernel_events) HIPCHK(p, hipEventRecord(c.ev[3 + 5 * it], st));
  }
  if (do_chord) {
    if (p->per_kernel_events) HIPCHK(p, hipEventRecord(c.ev[4 + 5 * it], cs));
    hipLaunchKernelGGL(p->chord_fn, dim3(c.B), dim3(KTC), chord_lds_bytes(p->S.n_stages, p->dp.sw_on ? p->dp.sw_steps : 0), cs, D, c.W, c.B);
    if (p->per_kernel_events) HIPCHK(p, hipEventRecord(c.ev[5 + 5 * it], cs));
  }
  if (fork) {
    HIPCHK(p, hipEventRecord(c.ev_join, cs));
    HIPCHK(p, hipStreamWaitEvent(st, c.ev_join, 0));
  }
  hipLaunchKernelGGL(k_step, dim3(c.B), dim3(ET), p->eval_lds, st, D, c.W, c.B, it, kinds);
  if (post) hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, st, c.W.n_active, c.h_active_dev + 4 * it, p->seq);
  if (p->per_kernel_events) HIPCHK(p, hipEventRecord(c.ev[6 + 5 * it], st));
  return 0;
}

// The part [b0, b0 + B) of the planner's workspace and of the caller's buffers, as a workspace of its own (every array is
// indexed by the problem's number inside its launch).  Strides: the allocations in qtos_planner_create.
static DevWork work_slice(const QtosPlanner *p, const DevWork &w, int b0, int *n_active) {
  DevWork s = w;
  const size_t n = p->M.n_vars, m = p->M.n_cons, ns = p->M.n_sol, NS = p->S.n_stages, b = (size_t)b0;
  auto off = [&](auto *&ptr, size_t stride) { if (ptr) ptr += b * stride; };
  off(s.start, QTOS_START_DOUBLES); off(s.goal, 3); off(s.warm, n); off(s.map_id, 1);
  off(s.x, n); off(s.g, m); off(s.gt, m); off(s.s, m); off(s.zl, m); off(s.zu, m); off(s.ds, m); off(s.dzl, m); off(s.dzu, m);
  off(s.sig, m); off(s.w, m); off(s.panel, (size_t)p->dp.panel_stride); off(s.dx, ns); off(s.stream, (size_t)p->dp.stream_len);
  off(s.mu, 1); off(s.viol, 1); off(s.trace, ((size_t)p->dp.max_iter + 1) * 4); off(s.best_viol, 1); off(s.xbest, n);
  off(s.held, 1); off(s.best_it, 1); off(s.status, 1); off(s.iters, 1); off(s.done, 1);
  off(s.chord, 1); off(s.chord_run, 1); off(s.jam, 1); off(s.rhs, (size_t)p->S.n_unknowns); off(s.minv, NS * PIV * PIV);
  off(s.sol, NS * PIV); off(s.sol0, NS * PIV); off(s.dx0, ns); off(s.ur, m);
  off(s.nodes_out, n); off(s.viol_out, 1); off(s.status_out, 1); off(s.iters_out, 1);
  s.n_active = n_active;
  return s;
}

int qtos_plan_submit(QtosPlanner *p, int B, const double *d_start, const double *d_goal,
                     const int *d_map_id, const double *d_warm, double *d_nodes_out,
                     int *d_status_out, int *d_iters_out, double *d_viol_out, void *stream_) {
  if (!p || B < 1 || B > p->max_batch || !d_start || !d_goal || !d_nodes_out) return -1;
  int expected = 0;
  if (!p->busy.compare_exchange_strong(expected, 1)) { p->err = "the planner handle already serves a call (one call per handle at a time)"; return -5; }
  auto fail = [&](int rc) { p->call_open = false; for (auto &L : p->lanes) L.open = false; p->busy.store(0); return rc; };
  hipStream_t st = (hipStream_t)stream_;
  if (hipSetDevice(p->device) != hipSuccess) return fail(-2);
  p->call_open = true;
  p->call_stream = st;
  // 16 bits of sequence number in the count word (0xffff = the reset pattern of the slots): the number wraps after 65 534 calls
  // of the handle.  A stale word can only be taken for this call's if a blind launch of the call 65 534 submits earlier were
  // still to store -- but a call is submitted only after the previous one was collected (p->busy above) and a call queues at
  // most max_iter blind launches, each a few microseconds of work once its problems are done: the launches of call n - 65 534
  // ran tens of thousands of collected calls ago.  Blind iterations are off by default (spec_cap 1).
  p->seq = p->seq >= 0xfffeu ? 1u : p->seq + 1u;
  DevWork W = p->wk;
  W.start = d_start; W.goal = d_goal; W.map_id = d_map_id; W.warm = d_warm;
  // (the kernel that finishes a problem writes its result: kernels.hpp export_problem)
  W.nodes_out = d_nodes_out; W.status_out = d_status_out; W.iters_out = d_iters_out; W.viol_out = d_viol_out;
  W.totals = (unsigned long long *)p->d_totals;
  const DevPlan &D = p->dp;
#define SUBCHK(call_) do { hipError_t e_ = (call_); if (e_ != hipSuccess) { p->err = std::string(#call_) + ": " + hipGetErrorString(e_); return fail(e_ == hipErrorOutOfMemory ? -3 : -2); } } while (0)
  // parts: one lane up to the GPU's compute units, then as many lanes as there are (equal parts)
  const int n_used = B <= p->lane_chunk ? 1 : std::min((int)p->lanes.size(), (B + p->lane_chunk - 1) / p->lane_chunk);
  const int per = (B + n_used - 1) / n_used;
  p->lanes_used = n_used;
  if (n_used > 1) SUBCHK(hipEventRecord(p->ev_in, st));
  const int ev_start = 2 + 5 * p->max_slots;
  // the launch pattern serves calls of one lane (a call cut into lanes keeps the informed loop per lane)
  const bool by_pattern = p->spec_pattern && n_used == 1 && !p->pat.empty() && p->spec_cap <= 1;
  if (by_pattern) p->n_pattern_calls++;
  for (int j = 0; j < n_used; ++j) {
    QtosPlanner::Lane &c = p->lanes[j];
    c.open = true; c.b0 = j * per; c.B = std::min(per, B - c.b0); c.spec = c.enq = c.chk = 0; c.spins = 0; c.n_informed = 0;
    c.by_pattern = by_pattern;
    c.st = j == 0 ? st : c.own;
    c.W = work_slice(p, W, c.b0, c.d_n_active);
    if (j > 0) SUBCHK(hipStreamWaitEvent(c.st, p->ev_in, 0));
    std::memset(c.h_active, 0xff, 4 * sizeof(int) * ((size_t)p->max_slots + 1));   // no counts yet (late stores of an earlier call carry another sequence number)
    SUBCHK(hipMemsetAsync(c.W.n_active, 0, 4 * sizeof(int), c.st));
    SUBCHK(hipEventRecord(c.ev[0], c.st));
    hipLaunchKernelGGL(k_start, dim3(c.B), dim3(ET), p->eval_lds, c.st, D, c.W, c.B);
    // counts after k_start (slot max_slots of the pinned array), event ev_start; a call that follows the pattern reads no
    // counts in front of the pattern's last slot and posts none
    if (!by_pattern) hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, c.st, c.W.n_active, c.h_active_dev + 4 * p->max_slots, p->seq);
    if (p->per_kernel_events) SUBCHK(hipEventRecord(c.ev[ev_start], c.st));
    if (by_pattern) {
      // the prefix of launch slots the handle's last two calls agree on, queued at once
      c.spec = std::min((int)p->pat.size(), p->max_slots);
      for (int it = 0; it < c.spec; ++it)
        if (int rc = queue_iteration(p, c, it, p->pat[it], it + 1 == c.spec)) return fail(rc);
      c.enq = c.chk = c.spec;     // (the first counts the host reads are those behind slot spec - 1)
    } else {
      // Blind iterations: as many as the previous call of this handle needed (its slowest problem), queued without a host
      // round trip (qtos_set_speculation; 1 by default: the first iteration).  A batch that needs more is continued by
      // qtos_plan_poll from the counts the iterations send back; blind launches behind the end find every problem done.
      c.spec = std::max(0, std::min(std::min(p->spec_next, p->spec_cap), D.max_iter));
      for (int it = 0; it < c.spec; ++it)
        if (int rc = queue_iteration(p, c, it, it >= 1 ? 3 : 1, true)) return fail(rc);
      c.enq = c.spec;
    }
  }
#undef SUBCHK
  if (n_used == 1 && hipEventRecord(p->lanes[0].ev[1], st) != hipSuccess) return fail(-2);
  p->last_stream = st;
  return 0;
}

// One lane as far as the counts that have arrived allow.  *lane_done: every problem of the part has handed its result over.
static int advance_lane(QtosPlanner *p, QtosPlanner::Lane &c, bool *lane_done, bool *device_set) {
  *lane_done = false;
  for (;;) {
    // counts in front of slot c.chk (behind k_start / slot c.chk - 1): the word itself says when they are there
    // (k_post_counts stores into mapped host memory; qtos_plan_submit left -1 in every slot) -- an event query on top of
    // it is a driver call and waits for the end-of-kernel signal of the command processor
    const int *h = c.h_active + 4 * (c.chk == 0 ? p->max_slots : c.chk - 1);
    const unsigned long long v = __atomic_load_n((const unsigned long long *)h, __ATOMIC_ACQUIRE);
    if (v == ~0ull || (unsigned)(v >> 48) != p->seq) {   // nothing yet, or the late store of a blind launch of an earlier call
      // (a failed launch never fills the slot: look at the stream now and then)
      if ((++c.spins & 0xfffff) == 0) {
        const hipError_t q = hipStreamQuery(c.st);
        if (q != hipSuccess && q != hipErrorNotReady) { p->err = std::string("hipStreamQuery: ") + hipGetErrorString(q); return -2; }
      }
      return 0;
    }
    const int n = (int)(unsigned)(v & 0xffffffull), nc = (int)(unsigned)((v >> 24) & 0xffffffull);
    if (n <= 0 || c.chk >= p->max_slots) {
      // finished in front of slot c.chk (every problem has handed its result over: export_problem; a problem out of
      // iterations counts as finished)
      // (a call that followed the pattern learns here how many of its slots had work: h[3], the slots in which some problem
      //  took a step; the others read every slot's counts and stop at the first without work)
      const int slots = c.by_pattern ? std::max(0, std::min(h[3], c.chk)) : c.chk;
      for (int j = slots; j < c.enq; ++j) { c.was_kkt[j] = 0; c.was_chord[j] = 0; }   // blind launches behind the end: no work
      c.last_launches = slots;
      c.last_iters = slots;
      c.open = false;
      *lane_done = true;
      return 0;
    }
    if (c.chk < c.spec && !c.by_pattern) {
      // a blind iteration that did run: which of its two solve kernels had work
      c.was_kkt[c.chk] = n - nc > 0;
      c.was_chord[c.chk] = nc > 0 && p->chord_fn && c.chk >= 1;
    }
    if (c.chk == c.enq) {
      if (!*device_set) { HIPCHK(p, hipSetDevice(p->device)); *device_set = true; }
      if (int rc = queue_iteration(p, c, c.chk, (n - nc > 0 ? 1 : 0) | (nc > 0 && c.chk >= 1 ? 2 : 0), true)) return rc;
      c.enq++;
      c.n_informed++;
      if (p->lanes_used == 1) HIPCHK(p, hipEventRecord(c.ev[1], c.st));   // (the end-of-call event sits behind the last launch)
    }
    c.chk++;
  }
}

int qtos_plan_poll(QtosPlanner *p, int *done) {
  if (!p || !done) return -1;
  *done = 0;
  if (!p->call_open) { *done = 1; return 0; }
  bool device_set = false;     // (only in front of launches: hipSetDevice in a spin loop of several host threads is a lock fight)
  auto fail = [&](int rc) { p->call_open = false; for (auto &L : p->lanes) L.open = false; p->busy.store(0); return rc; };
  int n_open = 0;
  for (int j = 0; j < p->lanes_used; ++j) {
    QtosPlanner::Lane &c = p->lanes[j];
    if (!c.open) continue;
    bool lane_done = false;
    if (int rc = advance_lane(p, c, &lane_done, &device_set)) return fail(rc);
    if (!lane_done) { ++n_open; continue; }
    if (j > 0) {   // the caller's stream waits for the part that ran beside it
      if (!device_set) { if (hipSetDevice(p->device) != hipSuccess) return fail(-2); device_set = true; }
      if (hipEventRecord(c.ev_done, c.st) != hipSuccess || hipStreamWaitEvent(p->call_stream, c.ev_done, 0) != hipSuccess) return fail(-2);
    }
  }
  if (n_open) return 0;
  if (p->lanes_used > 1) {   // end of the call: behind every part
    if (!device_set && hipSetDevice(p->device) != hipSuccess) return fail(-2);
    if (hipEventRecord(p->lanes[0].ev[1], p->call_stream) != hipSuccess) return fail(-2);
  }
  int iters = 0;
  for (int j = 0; j < p->lanes_used; ++j) iters = std::max(iters, p->lanes[j].last_iters);
  p->spec_next = std::max(1, std::min(iters, p->dp.max_iter));
  if (p->lanes_used == 1) {
    // The launch pattern of the next call: the kinds this call's slots ran, as far as they agree with the call before it
    // (the first call of a handle is trusted as it is).  A problem that sat a slot of the pattern out (the code in the last
    // count words: (1 << 20) - slot) cuts this call's observation in front of that slot -- what the slots behind it ran
    // was the pattern's choice, not the batch's.
    const QtosPlanner::Lane &c = p->lanes[0];
    const int *h = c.h_active + 4 * (c.chk == 0 ? p->max_slots : c.chk - 1);
    int n_obs = c.last_launches;
    if (h[2] > 0) { n_obs = std::min(n_obs, (1 << 20) - h[2]); p->n_pattern_misses++; }
    std::vector<char> obs((size_t)std::max(n_obs, 0));
    for (int i = 0; i < n_obs; ++i) obs[i] = (char)((c.was_kkt[i] ? 1 : 0) | (c.was_chord[i] ? 2 : 0));
    size_t k = 0;
    if (p->obs_prev.empty()) k = obs.size();
    else while (k < obs.size() && k < p->obs_prev.size() && obs[k] == p->obs_prev[k]) ++k;
    p->pat.assign(obs.begin(), obs.begin() + k);
    p->obs_prev.swap(obs);
  }
  p->call_open = false;
  p->busy.store(0);
  *done = 1;
  return 0;
}

int qtos_plan_wait(QtosPlanner *p) {
  if (!p) return -1;
  int done = 0;
  while (!done)
    if (int rc = qtos_plan_poll(p, &done)) return rc;
  return 0;
}

int qtos_plan_batch_device(QtosPlanner *p, int B, const double *d_start, const double *d_goal,
                           const int *d_map_id, const double *d_warm, double *d_nodes_out,
                           int *d_status_out, int *d_iters_out, double *d_viol_out, void *stream_) {
  if (int rc = qtos_plan_submit(p, B, d_start, d_goal, d_map_id, d_warm, d_nodes_out, d_status_out, d_iters_out, d_viol_out, stream_)) return rc;
  return qtos_plan_wait(p);
}

int qtos_set_speculation(QtosPlanner *p, int max_blind_iterations) {
  if (!p || max_blind_iterations < 0) return -1;
  p->spec_cap = max_blind_iterations;
  return 0;
}

int qtos_plan_totals(QtosPlanner *p, long long *converged, long long *iterations, int reset) {
  if (!p) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  long long h[2] = {0, 0};
  // on the planner's own stream, behind the end event of the last call (the caller's stream may be gone by now)
  if (p->call_open) HIPCHK(p, qtos_plan_wait(p) ? hipErrorUnknown : hipSuccess);
  HIPCHK(p, hipStreamWaitEvent(p->own_stream, p->lanes[0].ev[1], 0));
  HIPCHK(p, hipMemcpyAsync(h, p->d_totals, sizeof(h), hipMemcpyDeviceToHost, p->own_stream));
  if (reset) HIPCHK(p, hipMemsetAsync(p->d_totals, 0, sizeof(h), p->own_stream));
  HIPCHK(p, hipStreamSynchronize(p->own_stream));
  if (converged) *converged = h[0];
  if (iterations) *iterations = h[1];
  return 0;
}

int qtos_plan_batch(QtosPlanner *p, int B, const double *start, const double *goal, const int *map_id,
                    const double *warm, double *nodes_out, int *status_out, int *iters_out, double *viol_out) {
  if (!p || B < 1 || B > p->max_batch || !start || !goal || !nodes_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  const size_t n = p->M.n_vars;
  if (map_id)   // (the kernels clamp too; a wrong id is the caller's bug and is reported)
    for (int b = 0; b < B; ++b)
      if (map_id[b] < 0 || map_id[b] >= std::max(p->dp.n_maps, 1)) { p->err = "map_id out of range"; return -1; }
  hipStream_t st = p->own_stream;   // stream-scoped: no device-wide synchronisation
  HIPCHK(p, hipMemcpyAsync(p->d_start, start, (size_t)B * QTOS_START_DOUBLES * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(p, hipMemcpyAsync(p->d_goal, goal, (size_t)B * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  if (map_id) HIPCHK(p, hipMemcpyAsync(p->d_map, map_id, B * sizeof(int), hipMemcpyHostToDevice, st));
  if (warm) HIPCHK(p, hipMemcpyAsync(p->d_warm, warm, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, st));
  int rc = qtos_plan_batch_device(p, B, p->d_start, p->d_goal, map_id ? p->d_map : nullptr,
                                  warm ? p->d_warm : nullptr, p->d_nodes, nullptr, nullptr, nullptr, (void *)st);
  if (rc) return rc;
  HIPCHK(p, hipMemcpyAsync(nodes_out, p->d_nodes, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, st));
  if (status_out) HIPCHK(p, hipMemcpyAsync(status_out, p->wk.status, B * sizeof(int), hipMemcpyDeviceToHost, st));
  if (iters_out) HIPCHK(p, hipMemcpyAsync(iters_out, p->wk.iters, B * sizeof(int), hipMemcpyDeviceToHost, st));
  if (viol_out) HIPCHK(p, hipMemcpyAsync(viol_out, p->wk.viol, B * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(p, hipStreamSynchronize(st));
  return 0;
}

// Timing of the last call from the HIP events of its lanes: KKT (chord) launches and their seconds summed over the lanes,
// total = first kernel of lane 0 to the end of the call, iterations = the slowest lane's.
int qtos_last_timing(QtosPlanner *p, double *kkt_seconds, int *kkt_launches, double *total_seconds, int *iterations) {
  if (!p) return -1;
  if (!p->per_kernel_events) { p->err = "per-kernel events are off (qtos_set_kernel_events)"; return -1; }
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipEventSynchronize(p->lanes[0].ev[1]));
  double kkt = 0;
  int n_kkt = 0, iters = 0;
  for (int j = 0; j < p->lanes_used; ++j) {
    const QtosPlanner::Lane &L = p->lanes[j];
    iters = std::max(iters, L.last_iters);
    for (int i = 0; i < L.last_launches; ++i) {
      if (!L.was_kkt[i]) continue;
      float ms = 0;
      HIPCHK(p, hipEventElapsedTime(&ms, L.ev[2 + 5 * i], L.ev[3 + 5 * i]));
      kkt += ms * 1e-3;
      ++n_kkt;
    }
  }
  float tot = 0;
  HIPCHK(p, hipEventElapsedTime(&tot, p->lanes[0].ev[0], p->lanes[0].ev[1]));
  if (kkt_seconds) *kkt_seconds = kkt;
  if (kkt_launches) *kkt_launches = n_kkt;
  if (total_seconds) *total_seconds = tot * 1e-3;
  if (iterations) *iterations = iters;
  return 0;
}

// Where the time of the last call went, from the same events (lane 0; one lane per call is what bench.py times):
//   out[0] first kernel -> end of the call            out[1] memset + k_start (+ its counts)
//   out[2] solve kernels (a slot's factorisation and chord solve side by side count once)
//   out[3] k_step + counts of every slot (from the end of the slot's solve to its last event)
//   out[4] GAPS: from a slot's last event to the first event of the next slot's solve -- the host reading the counts and
//          launching (zero between slots queued at submit time) -- plus whatever lies between the last slot and the end event
//   out[5] launch slots with work, out[6] slots queued at submit time (pattern / blind), out[7] launches that waited for the host
//   out[8] calls of the handle that followed a pattern so far, out[9] of those, calls in which a problem sat a slot out
//   n_out >= 14: out[10] seconds / out[11] launches of the factorising kernel, out[12] / out[13] of k_chord (qtos_last_timing's and
//   qtos_last_timing_chord's numbers: one call instead of three in a timed loop)
int qtos_last_timing_detail(QtosPlanner *p, double *out, int n_out) {
  if (!p || !out || n_out < 10) return -1;
  if (!p->per_kernel_events) { p->err = "per-kernel events are off (qtos_set_kernel_events)"; return -1; }
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipEventSynchronize(p->lanes[0].ev[1]));
  const QtosPlanner::Lane &L = p->lanes[0];
  const int ev_start = 2 + 5 * p->max_slots;
  auto ms = [&](hipEvent_t a, hipEvent_t b, double *d) { float t = 0; hipError_t e = hipEventElapsedTime(&t, a, b); *d = t * 1e-3; return e; };
  double tot = 0, start = 0, solve = 0, step = 0, gaps = 0, kkt_s = 0, chord_s = 0;
  int kkt_n = 0, chord_n = 0;
  HIPCHK(p, ms(L.ev[0], L.ev[1], &tot));
  HIPCHK(p, ms(L.ev[0], L.ev[ev_start], &start));
  hipEvent_t prev = L.ev[ev_start];
  for (int i = 0; i < L.enq; ++i) {
    if (!L.was_kkt[i] && !L.was_chord[i] && i >= L.last_launches) continue;   // (queued behind the end: nothing ran)
    hipEvent_t b0 = L.was_kkt[i] ? L.ev[2 + 5 * i] : (L.was_chord[i] ? L.ev[4 + 5 * i] : nullptr);
    double g = 0, sk = 0, sc = 0, st = 0;
    if (b0) {
      HIPCHK(p, ms(prev, b0, &g));
      if (L.was_kkt[i]) HIPCHK(p, ms(b0, L.ev[3 + 5 * i], &sk));
      if (L.was_chord[i]) HIPCHK(p, ms(b0, L.ev[5 + 5 * i], &sc));
      if (i < L.last_launches) {
        if (L.was_kkt[i]) { kkt_s += sk; ++kkt_n; }
        if (L.was_chord[i]) { double own = sc; if (L.was_kkt[i]) HIPCHK(p, ms(L.ev[4 + 5 * i], L.ev[5 + 5 * i], &own)); chord_s += own; ++chord_n; }
      }
      const double sv = std::max(sk, sc);
      double whole = 0;
      HIPCHK(p, ms(b0, L.ev[6 + 5 * i], &whole));
      st = std::max(0.0, whole - sv);
      gaps += std::max(0.0, g); solve += sv; step += st;
    } else {
      HIPCHK(p, ms(prev, L.ev[6 + 5 * i], &st));
      step += st;
    }
    prev = L.ev[6 + 5 * i];
  }
  gaps += std::max(0.0, tot - start - solve - step - gaps);
  out[0] = tot; out[1] = start; out[2] = solve; out[3] = step; out[4] = gaps;
  out[5] = L.last_launches; out[6] = L.spec; out[7] = L.n_informed;
  out[8] = (double)p->n_pattern_calls; out[9] = (double)p->n_pattern_misses;
  if (n_out >= 14) { out[10] = kkt_s; out[11] = kkt_n; out[12] = chord_s; out[13] = chord_n; }
  return 0;
}

int qtos_set_kernel_events(QtosPlanner *p, int on) {
  if (!p) return -1;
  p->per_kernel_events = on != 0;
  return 0;
}

int qtos_set_pattern_speculation(QtosPlanner *p, int on) {
  if (!p) return -1;
  p->spec_pattern = on != 0;
  if (!on) { p->pat.clear(); p->obs_prev.clear(); }
  return 0;
}

int qtos_env(const QtosPlanner *p, char *buf, int n) {
  if (!p || !buf || n < 1) return -1;
  const std::string d = p->env.describe();
  snprintf(buf, (size_t)n, "%s", d.c_str());
  return (int)d.size();
}

int qtos_last_timing_chord(QtosPlanner *p, double *chord_seconds, int *chord_launches) {
  if (!p) return -1;
  if (!p->per_kernel_events) { p->err = "per-kernel events are off (qtos_set_kernel_events)"; return -1; }
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipEventSynchronize(p->lanes[0].ev[1]));
  double t = 0;
  int nl = 0;
  for (int j = 0; j < p->lanes_used; ++j) {
    const QtosPlanner::Lane &L = p->lanes[j];
    for (int i = 0; i < L.last_launches; ++i) {
      if (!L.was_chord[i]) continue;
      float ms = 0;
      HIPCHK(p, hipEventElapsedTime(&ms, L.ev[4 + 5 * i], L.ev[5 + 5 * i]));
      t += ms * 1e-3;
      ++nl;
    }
  }
  if (chord_seconds) *chord_seconds = t;
  if (chord_launches) *chord_launches = nl;
  return 0;
}

int qtos_sample_csv_device(QtosPlanner *p, int B, const double *d_nodes, const double *d_t0, double hz,
                           int n_rows, double *d_rows_out, void *stream_) {
  if (!p || B < 1 || !d_nodes || !d_t0 || !d_rows_out || n_rows < 1 || !(hz > 0)) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  dim3 grid((n_rows + 255) / 256, B);
  hipLaunchKernelGGL(k_sample, grid, dim3(256), 0, (hipStream_t)stream_, p->sp, d_nodes, d_t0, hz, n_rows, d_rows_out, B);
  HIPCHK(p, hipGetLastError());
  return 0;
}

// The plan as the text file the reference fetches from its container (build/traj.csv -> ./data/traj/towr.csv: scripts/main.py:90-92):
// n_rows x 37 numbers printed like the solver's C++ stream prints them ("%g"), csv_writer.hpp.  Host only, no planner handle.
int qtos_write_csv(const char *path, const double *rows, int n_rows, int n_threads) {
  return write_csv_file(path, rows, n_rows, QTOS_CSV_COLS, n_threads);
}

int qtos_sample_csv(QtosPlanner *p, int B, const double *nodes, const double *t0, double hz, int n_rows, double *rows_out) {
  if (!p || B < 1 || !nodes || !t0 || !rows_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  double *d_nodes = nullptr, *d_t0 = nullptr, *d_rows = nullptr;
  const size_t n = p->M.n_vars, rows_bytes = (size_t)B * n_rows * QTOS_CSV_COLS * sizeof(double);
  HIPCHK(p, hipMalloc((void **)&d_nodes, (size_t)B * n * sizeof(double)));
  HIPCHK(p, hipMalloc((void **)&d_t0, B * sizeof(double)));
  HIPCHK(p, hipMalloc((void **)&d_rows, rows_bytes));
  HIPCHK(p, hipMemcpy(d_nodes, nodes, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(d_t0, t0, B * sizeof(double), hipMemcpyHostToDevice));
  int rc = qtos_sample_csv_device(p, B, d_nodes, d_t0, hz, n_rows, d_rows, nullptr);
  if (!rc) {
    hipError_t e = hipMemcpy(rows_out, d_rows, rows_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = -2;
  }
  (void)hipFree(d_nodes); (void)hipFree(d_t0); (void)hipFree(d_rows);
  return rc;
}

// ---- time-shifted warm start (receding-horizon replans) ---------------------------------------------
// Every variable of the new plan sits at a node time t; where the previous plan still covers offset + t its
// spline is evaluated there (positions / velocities / forces in the common world frame), beyond its horizon
// the straight-line guess of the new problem takes over; fixed variables carry the new start / goal.
__global__ __launch_bounds__(256) void k_shift_warm(DevPlan P, SamplePlan S, const double *prev, const double *offset,
                                                    const double *start, const double *goal, const int *map_id, double *out, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const double *x = prev + (size_t)b * P.n_vars, *st = start + (size_t)b * QTOS_START_DOUBLES, *gl = goal + (size_t)b * 3;
  const int map = map_id ? map_id[b] : 0;
  const double off = offset[b];
  for (int v = threadIdx.x; v < P.n_vars; v += blockDim.x) {
    const InitDesc I = P.init[v];
    double val;
    if (I.fix_src >= 0) val = I.fix_src < 24 ? st[I.fix_src] : (I.fix_src < 26 ? gl[I.fix_src - 24] : 0.0);
    else {
      const double t = off + P.var_time[v];
      if (t <= S.T + 1e-9) {
        const SampleSpline &sp = I.set == 0 ? S.lin : (I.set == 1 ? S.ang : (I.set < 6 ? S.eem[I.set - 2] : S.eef[I.set - 6]));
        double o3[3];
        sample_spline(sp, x, fmin(t, S.T), I.is_vel, o3);
        val = o3[I.dim];
      } else val = straight_line_value(P, I, st, gl, map);
    }
    out[(size_t)b * P.n_vars + v] = val;
  }
}

int qtos_shift_warm_device(QtosPlanner *p, int B, const double *d_nodes_prev, const double *d_offset, const double *d_start,
                           const double *d_goal, const int *d_map_id, double *d_warm_out, void *stream_) {
  if (!p || B < 1 || !d_nodes_prev || !d_offset || !d_start || !d_goal || !d_warm_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  hipLaunchKernelGGL(k_shift_warm, dim3(B), dim3(256), 0, (hipStream_t)stream_, p->dp, p->sp, d_nodes_prev, d_offset, d_start, d_goal,
                     d_map_id, d_warm_out, B);
  HIPCHK(p, hipGetLastError());
  return 0;
}

int qtos_shift_warm(QtosPlanner *p, int B, const double *nodes_prev, const double *offset, const double *start, const double *goal,
                    const int *map_id, double *warm_out) {
  if (!p || B < 1 || B > p->max_batch || !nodes_prev || !offset || !start || !goal || !warm_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  const size_t n = p->M.n_vars;
  hipStream_t st = p->own_stream;
  double *d_off = nullptr;
  HIPCHK(p, hipMalloc((void **)&d_off, B * sizeof(double)));
  hipError_t e = hipSuccess;
  auto cp = [&](void *dst, const void *src, size_t bytes, hipMemcpyKind k) { if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, k, st); };
  cp(p->d_nodes, nodes_prev, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice);
  cp(d_off, offset, B * sizeof(double), hipMemcpyHostToDevice);
  cp(p->d_start, start, (size_t)B * QTOS_START_DOUBLES * sizeof(double), hipMemcpyHostToDevice);
  cp(p->d_goal, goal, (size_t)B * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (map_id) cp(p->d_map, map_id, B * sizeof(int), hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? qtos_shift_warm_device(p, B, p->d_nodes, d_off, p->d_start, p->d_goal, map_id ? p->d_map : nullptr, p->d_warm, (void *)st) : -2;
  if (!rc) {
    cp(warm_out, p->d_warm, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = -2;
  }
  (void)hipFree(d_off);
  return rc;
}

int qtos_set_init_table(QtosPlanner *p, int ndx, const double *dx, int ndy, const double *dy, const double *nodes) {
  if (!p) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  for (void *q : {(void *)p->d_table, (void *)p->d_tab_dx, (void *)p->d_tab_dy})
    if (q) (void)hipFree(q);
  p->d_table = p->d_tab_dx = p->d_tab_dy = nullptr;
  p->dp.table = p->dp.tab_dx = p->dp.tab_dy = nullptr;
  p->dp.tab_ndx = p->dp.tab_ndy = 0;
  if (ndx <= 0 || ndy <= 0 || !dx || !dy || !nodes) return 0;
  for (int i = 1; i < ndx; ++i) if (!(dx[i] > dx[i - 1])) return -1;
  for (int j = 1; j < ndy; ++j) if (!(dy[j] > dy[j - 1])) return -1;
  const size_t cnt = (size_t)ndx * ndy * p->M.n_vars;
  HIPCHK(p, hipMalloc((void **)&p->d_table, cnt * sizeof(double)));
  HIPCHK(p, hipMalloc((void **)&p->d_tab_dx, ndx * sizeof(double)));
  HIPCHK(p, hipMalloc((void **)&p->d_tab_dy, ndy * sizeof(double)));
  HIPCHK(p, hipMemcpy(p->d_table, nodes, cnt * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(p->d_tab_dx, dx, ndx * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(p->d_tab_dy, dy, ndy * sizeof(double), hipMemcpyHostToDevice));
  p->dp.table = p->d_table; p->dp.tab_dx = p->d_tab_dx; p->dp.tab_dy = p->d_tab_dy;
  p->dp.tab_ndx = ndx; p->dp.tab_ndy = ndy;
  return 0;
}

// ---- introspection -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_debug_guess(DevPlan P, DevWork W, int B, double *out) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const double *st = W.start + (size_t)b * QTOS_START_DOUBLES, *gl = W.goal + (size_t)b * 3;
  const int map = W.map_id ? W.map_id[b] : 0;
  const TableCell tc = table_cell(P, st, gl);
  for (int v = threadIdx.x; v < P.n_vars; v += blockDim.x) out[(size_t)b * P.n_vars + v] = initial_value(P, W, b, v, st, gl, map, tc);
  __syncthreads();
  double *y = out + (size_t)b * P.n_vars;
  for (int i = threadIdx.x; i < P.n_psw; i += blockDim.x)   // (reduced swings: k_start places the mid nodes on the swing rule)
    y[P.psw_var[i]] = fma(P.psw_w[2 * i], y[P.psw_src[2 * i]], P.psw_w[2 * i + 1] * y[P.psw_src[2 * i + 1]]);
}

// reduced base: what given nodes become when a solve starts from them (k_start's projection onto the coefficients' space)
__global__ __launch_bounds__(256) void k_project_nodes(DevPlan P, const double *in, double *out, int B) {
  extern __shared__ double cf[];
  const int b = blockIdx.x, n = P.n_vars;
  if (b >= B) return;
  const double *x = in + (size_t)b * n;
  double *y = out + (size_t)b * n;
  for (int v = threadIdx.x; v < n; v += blockDim.x) y[v] = x[v];
  for (int i = threadIdx.x; i < P.n_psw; i += blockDim.x)   // (reduced swings: mid nodes onto the swing rule; their sources are footholds)
    y[P.psw_var[i]] = fma(P.psw_w[2 * i], x[P.psw_src[2 * i]], P.psw_w[2 * i + 1] * x[P.psw_src[2 * i + 1]]);
  for (int c = threadIdx.x; c < P.n_coef; c += blockDim.x) {
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc = fma(P.pc_w[4 * c + a], x[P.pc_var[4 * c + a]], acc);
    cf[c] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P.n_pz; i += blockDim.x) {
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc = fma(P.pz_w[4 * i + a], cf[P.pz_col[4 * i + a]], acc);
    y[P.pz_var[i]] = acc;
  }
}

int qtos_project_nodes(QtosPlanner *p, int B, const double *nodes, double *nodes_out) {
  if (!p || B < 1 || B > p->max_batch || !nodes || !nodes_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  const size_t bytes = (size_t)B * p->M.n_vars * sizeof(double);
  HIPCHK(p, hipMemcpy(p->d_warm, nodes, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_project_nodes, dim3(B), dim3(256), sizeof(double) * std::max(p->dp.n_coef, 1), 0, p->dp, p->d_warm, p->d_nodes, B);
  HIPCHK(p, hipDeviceSynchronize());
  HIPCHK(p, hipMemcpy(nodes_out, p->d_nodes, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int qtos_debug_initial_guess(QtosPlanner *p, int B, const double *start, const double *goal, const int *map_id, double *nodes_out) {
  if (!p || B < 1 || B > p->max_batch || !start || !goal || !nodes_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipMemcpy(p->d_start, start, (size_t)B * QTOS_START_DOUBLES * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(p->d_goal, goal, (size_t)B * 3 * sizeof(double), hipMemcpyHostToDevice));
  if (map_id) HIPCHK(p, hipMemcpy(p->d_map, map_id, (size_t)B * sizeof(int), hipMemcpyHostToDevice));
  DevWork W = p->wk;
  W.start = p->d_start; W.goal = p->d_goal; W.map_id = map_id ? p->d_map : nullptr; W.warm = nullptr;
  hipLaunchKernelGGL(k_debug_guess, dim3(B), dim3(256), 0, 0, p->dp, W, B, p->d_nodes);
  HIPCHK(p, hipDeviceSynchronize());
  HIPCHK(p, hipMemcpy(nodes_out, p->d_nodes, (size_t)B * p->M.n_vars * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

__global__ __launch_bounds__(256) void k_debug_eval(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int n = P.n_vars, m = P.n_cons;
  double *x = W.x + (size_t)b * n, *g = W.g + (size_t)b * m;
  const double *st = W.start + (size_t)b * QTOS_START_DOUBLES, *gl = W.goal + (size_t)b * 3;
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    const InitDesc I = P.init[v];
    x[v] = I.fix_src >= 0 ? (I.fix_src < 24 ? st[I.fix_src] : (I.fix_src < 26 ? gl[I.fix_src - 24] : 0.0))
                          : W.warm[(size_t)b * n + v];
  }
  __syncthreads();
  extern __shared__ double evl[];
  eval_all<true>(P, W.map_id ? W.map_id[b] : 0, x, g, W.stream + (size_t)b * P.stream_len, evl);
  if (threadIdx.x == 0) W.done[b] = 0;
}

__global__ __launch_bounds__(256) void k_debug_pack(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const size_t m = P.n_cons;
  const double *g = W.g + b * m, *sig = W.sig + b * m, *w = W.w + b * m;
  double *stream = W.stream + (size_t)b * P.stream_len;
  for (int r = threadIdx.x; r < P.n_cons; r += blockDim.x) {
    if (P.row_kind[r] == 1) stream[P.rhs_pos[r]] = -g[r];
    if (P.row_kind[r] == 2) { stream[P.sig_pos[r]] = sig[r]; stream[P.w_pos[r]] = w[r]; }
  }
}

// reduced base: the step of the base node values from the step of the coefficients (what k_step does at its start), for
// the entry points that hand dx out
__global__ __launch_bounds__(256) void k_recover_dx(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  double *dx = W.dx + (size_t)b * P.n_sol;
  for (int i = threadIdx.x; i < P.n_rec; i += blockDim.x) {
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) {
      const int col = P.rec_col[4 * i + a];
      if (col >= 0) acc = fma(P.rec_w[4 * i + a], dx[col], acc);
    }
    dx[P.rec_var[i]] = acc;
  }
}
// dx (node space, B x n_vars) of the last debug solve to the host
static int copy_dx_out(QtosPlanner *p, const DevWork &W, int B, double *dx_out) {
  if (p->dp.n_rec) hipLaunchKernelGGL(k_recover_dx, dim3(B), dim3(256), 0, 0, p->dp, W, B);
  HIPCHK(p, hipDeviceSynchronize());
  HIPCHK(p, hipMemcpy2D(dx_out, p->M.n_vars * sizeof(double), W.dx, p->M.n_sol * sizeof(double), p->M.n_vars * sizeof(double), B, hipMemcpyDeviceToHost));
  return 0;
}

static int debug_upload(QtosPlanner *p, int B, const double *start, const double *goal, const int *map_id, const double *nodes, DevWork *W) {
  if (!p || B < 1 || B > p->max_batch || !start || !goal || !nodes) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  const size_t n = p->M.n_vars;
  HIPCHK(p, hipMemcpy(p->d_start, start, (size_t)B * QTOS_START_DOUBLES * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(p->d_goal, goal, (size_t)B * 3 * sizeof(double), hipMemcpyHostToDevice));
  if (map_id) HIPCHK(p, hipMemcpy(p->d_map, map_id, B * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(p->d_warm, nodes, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice));
  *W = p->wk;
  W->start = p->d_start; W->goal = p->d_goal; W->map_id = map_id ? p->d_map : nullptr; W->warm = p->d_warm;
  hipLaunchKernelGGL(k_debug_eval, dim3(B), dim3(256), p->eval_lds, 0, p->dp, *W, B);
  HIPCHK(p, hipDeviceSynchronize());
  return 0;
}

int qtos_debug_eval(QtosPlanner *p, int B, const double *start, const double *goal, const int *map_id,
                    const double *nodes, double *g_out, double *J_out) {
  if (J_out && p && p->M.reduce_base) { p->err = "qtos_debug_eval: the Jacobian in node space is not formed with reduce_base (create the planner without it)"; return -1; }
  DevWork W;
  int rc = debug_upload(p, B, start, goal, map_id, nodes, &W);
  if (rc) return rc;
  const HostModel &M = p->M;
  const size_t n = M.n_vars, m = M.n_cons;
  if (g_out) HIPCHK(p, hipMemcpy(g_out, W.g, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost));
  if (J_out) {
    const size_t SL = p->S.pack_src.size();
    std::vector<double> G((size_t)B * SL);
    HIPCHK(p, hipMemcpy(G.data(), W.stream, G.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::memset(J_out, 0, (size_t)B * m * n * sizeof(double));
    for (int b = 0; b < B; ++b)
      for (const Block &blk : M.blocks)
        for (int r = 0; r < blk.m; ++r)
          for (int a = 0; a < blk.n; ++a) {
            double v;
            if (blk.gstatic) v = M.g_static[blk.goff + r * blk.n + a];
            else if (blk.kind == 1) v = G[(size_t)b * SL + blk.goff + r * blk.n + a];
            else v = G[(size_t)b * SL + p->S.eq_pos[blk.goff + r * blk.n + a]];
            J_out[((size_t)b * m + blk.row0 + r) * n + M.block_cols[blk.col_off + a]] = v;
          }
  }
  return 0;
}

int qtos_debug_newton(QtosPlanner *p, int B, const double *start, const double *goal, const int *map_id,
                      const double *nodes, const double *sig, const double *w, double *dx_out) {
  DevWork W;
  int rc = debug_upload(p, B, start, goal, map_id, nodes, &W);
  if (rc) return rc;
  const size_t n = p->M.n_vars, m = p->M.n_cons;
  HIPCHK(p, hipMemcpy(W.sig, sig, (size_t)B * m * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(p, hipMemcpy(W.w, w, (size_t)B * m * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_pack, dim3(B), dim3(256), 0, 0, p->dp, W, B);
  hipLaunchKernelGGL(p->kkt_fn, dim3(B), dim3(p->kkt_threads), p->kkt_lds, 0, p->dp, W, B);
  HIPCHK(p, hipDeviceSynchronize());
  if (int rc = copy_dx_out(p, W, B, dx_out)) return rc;
  return 0;
}

// right-hand side of the KKT system from the state debug_upload / k_debug_pack left behind (k_step's formula)
__global__ __launch_bounds__(256) void k_debug_rhs(DevPlan P, DevWork W, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const size_t m = P.n_cons;
  const double *g = W.g + b * m, *wr = W.w + b * m, *Gs = W.stream + (size_t)b * P.stream_len;
  double *rhs = W.rhs + (size_t)b * P.n_unknowns;
  for (int p = threadIdx.x; p < P.n_unknowns; p += blockDim.x) {
    const int t0 = P.rhs_ptr[p], t1 = P.rhs_ptr[p + 1];
    double acc = 0.0;
    if (t1 - t0 == 1 && P.rhs_gpos[t0] < 0) acc = -g[P.rhs_row[t0]];
    else
      for (int t = t0; t < t1; ++t) acc = fma(-Gs[P.rhs_gpos[t]], wr[P.rhs_row[t]], acc);
    rhs[p] = acc;
  }
  if (threadIdx.x == 0) { W.chord[b] = 1; atomicAdd(W.n_active + 1, 1); }
}
__global__ void k_debug_flag_chord(DevWork W, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { W.chord[b] = 1; atomicAdd(W.n_active + 1, 1); }
}
__global__ void k_debug_unchord(DevWork W, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) W.chord[b] = 0;
  if (b == 0) W.n_active[1] = 0;
}

/* the same solve as qtos_debug_newton once more, this time by k_chord: the factorisation the preceding
 * qtos_debug_newton call left behind + the right-hand side in elimination order.  dx_out: B x n_vars */
int qtos_debug_chord(QtosPlanner *p, int B, double *dx_out) {
  if (!p || B < 1 || B > p->max_batch || !dx_out || !p->chord_fn) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  DevWork W = p->wk;
  hipLaunchKernelGGL(k_debug_rhs, dim3(B), dim3(256), 0, 0, p->dp, W, B);
  hipLaunchKernelGGL(p->chord_fn, dim3(B), dim3(KTC), chord_lds_bytes(p->S.n_stages, p->dp.sw_on ? p->dp.sw_steps : 0), 0, p->dp, W, B);
  hipLaunchKernelGGL(k_debug_unchord, dim3((B + 63) / 64), dim3(64), 0, 0, W, B);
  HIPCHK(p, hipDeviceSynchronize());
  if (int rc = copy_dx_out(p, W, B, dx_out)) return rc;
  return 0;
}

/* A-posteriori check of the system the preceding qtos_debug_newton call solved: res_rel_out[b] = max |b - K x| / max |b|
 * with K applied from the problem's stream (k_residual: no use of the factorisation).  refine != 0: one step of
 * iterative refinement through the stored factorisation first -- r = b - K x, K e = r by k_chord, x += e -- and the
 * residual of the refined solution; dx_out (B x n_vars, may be NULL) receives the solution. */
int qtos_debug_residual(QtosPlanner *p, int B, int refine, double *dx_out, double *res_rel_out) {
  if (!p || B < 1 || B > p->max_batch || !res_rel_out) return -1;
  if (refine && !p->chord_fn) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  DevWork W = p->wk;
  double *d_out = nullptr;
  HIPCHK(p, hipMalloc((void **)&d_out, B * sizeof(double)));
  if (refine) {
    hipLaunchKernelGGL(k_residual, dim3(B), dim3(512), 0, 0, p->dp, W, B, (double *)nullptr, 1);   // r -> W.rhs, x remembered
    hipLaunchKernelGGL(k_debug_flag_chord, dim3((B + 63) / 64), dim3(64), 0, 0, W, B);
    hipLaunchKernelGGL(p->chord_fn, dim3(B), dim3(KTC), chord_lds_bytes(p->S.n_stages, p->dp.sw_on ? p->dp.sw_steps : 0), 0, p->dp, W, B);
    hipLaunchKernelGGL(k_debug_unchord, dim3((B + 63) / 64), dim3(64), 0, 0, W, B);
    hipLaunchKernelGGL(k_refine_add, dim3(B), dim3(512), 0, 0, p->dp, W, B);
  }
  hipLaunchKernelGGL(k_residual, dim3(B), dim3(512), 0, 0, p->dp, W, B, d_out, 0);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(res_rel_out, d_out, B * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && dx_out && copy_dx_out(p, W, B, dx_out)) e = hipErrorUnknown;
  (void)hipFree(d_out);
  if (e != hipSuccess) { p->err = std::string("qtos_debug_residual: ") + hipGetErrorString(e); return -2; }
  return 0;
}

int qtos_build_flags(void) {
  int f = 0;
#ifdef QTOS_EXPERIMENTS
  f |= 1;
#endif
#ifdef QTOS_STAMPS
  f |= 2;
#endif
#ifdef QTOS_DEV_F128
  f |= 4;
#endif
  return f;
}
int qtos_kkt_kernel(const QtosPlanner *p, char *buf, int n) {
  if (!p) return -1;
  char name[64];
  const int F = p->S.front;
  if (p->use_kkt5) snprintf(name, sizeof name, "k_kkt5<%d>", F);
  else if (p->use_kkt3) snprintf(name, sizeof name, "k_kkt3<%d, 1>", F);
  else snprintf(name, sizeof name, "k_kkt2<%d%s>", F, p->dp.n_cont > 0 ? ", true" : "");
  if (buf && n > 0) snprintf(buf, (size_t)n, "%s", name);
  return (int)strlen(name);
}
/* diagnostic: the stage stream of problem b (qtos_debug_stream_len doubles) */
int qtos_debug_stream_len(const QtosPlanner *p) { return p ? (int)p->S.pack_src.size() : -1; }
int qtos_debug_read_stream(QtosPlanner *p, int b, double *out) {
  if (!p || !out || b < 0 || b >= p->max_batch) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipMemcpy(out, p->wk.stream + (size_t)b * p->S.pack_src.size(), p->S.pack_src.size() * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
/* diagnostic: the vector W.rhs of problem b by position of the elimination order (n_stages x 16 doubles; S.n_unknowns counts
 * positions, dummy pivots included): after qtos_debug_residual the residual */
int qtos_debug_read_rhs(QtosPlanner *p, int b, double *out) {
  if (!p || !out || b < 0 || b >= p->max_batch) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  HIPCHK(p, hipMemcpy(out, p->wk.rhs + (size_t)b * p->S.n_unknowns, p->S.n_unknowns * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int qtos_debug_structure(const QtosPlanner *p, int *row_kind, int *var_free, int *order) {
  if (!p) return -1;
  if (row_kind) std::memcpy(row_kind, p->M.row_kind.data(), p->M.n_cons * sizeof(int));
  if (var_free)
    for (int v = 0; v < p->M.n_vars; ++v) var_free[v] = p->M.is_free(v) ? 1 : 0;
  if (order) {   // n_stages x 16 positions: the unknown eliminated there (variable, or n_vars + row), -1 = dummy pivot
    std::memcpy(order, p->S.order.data(), p->S.n_unknowns * sizeof(int));
    for (int i = p->S.n_unknowns; i < p->S.n_stages * PIV; ++i) order[i] = -1;
  }
  return 0;
}

int qtos_debug_factor(QtosPlanner *p, int b, double *panel_out, int *piv_slot_out) {
  if (!p || b < 0 || b >= p->max_batch) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  if (panel_out)
    HIPCHK(p, hipMemcpy(panel_out, p->wk.panel + (size_t)b * p->dp.panel_stride, (size_t)p->dp.panel_stride * sizeof(double), hipMemcpyDeviceToHost));
  if (panel_out) {   // rows the kernel does not store (dead slots, the stage's own pivots) are zero by definition
    const int F = p->S.front, NS = p->S.n_stages;
    for (int k = 0; k < NS; ++k)
      for (int r = 0; r < F; ++r)
        if (!((p->S.amask[(size_t)k * 8 + (r >> 5)] >> (r & 31)) & 1u))
          std::memset(panel_out + ((size_t)k * (F + 1) + 1 + r) * PIV, 0, PIV * sizeof(double));
  }
  if (piv_slot_out) std::memcpy(piv_slot_out, p->S.piv_slot.data(), p->S.piv_slot.size() * sizeof(int));
  return 0;
}

int qtos_debug_trace(QtosPlanner *p, int b, double *trace_out) {
  if (!p || b < 0 || b >= p->max_batch || !trace_out) return -1;
  HIPCHK(p, hipSetDevice(p->device));
  int iters = 0;
  HIPCHK(p, hipMemcpy(&iters, p->wk.iters + b, sizeof(int), hipMemcpyDeviceToHost));
  const int rows = iters + 1, stride = p->M.P.max_iter + 1;
  // the caller's buffer holds max_iter + 1 rows (diagnostic builds park phase stamps past `rows`)
  HIPCHK(p, hipMemcpy(trace_out, p->wk.trace + (size_t)b * stride * 4, (size_t)stride * 4 * sizeof(double), hipMemcpyDeviceToHost));
  return rows;
}

}  // extern "C"
