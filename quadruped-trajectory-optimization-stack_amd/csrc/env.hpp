// env.hpp -- every environment variable the library reads, parsed in ONE place.
//
// The header promises "no global state": the environment is read once per entry point that builds a planner or an analysis
// (qtos_planner_create, qtos_analyze*), into this struct; the planner keeps its copy for its whole life and qtos_env() hands
// the caller the values the handle actually runs with.  Nothing else in the library calls getenv.  All of these are diagnostics
// or measured alternatives; the defaults are the measured optimum (include/qtos_planner.h lists what each one does).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>

namespace qtos {

struct QtosEnv {
  int kkt = 0;            // QTOS_KKT            0 = the default choice of the factor + solve kernel; 2 | 4 | 6 force one
  int lanes = 1;          // QTOS_LANES          parts a call larger than the GPU is cut into (1 .. 4)
  int short_stages = -1;  // QTOS_SHORT_STAGES   -1 unset (where they cost no stage / above 128 slots), 0 never, 1 always
                          //                     (QTOS_NO_SHORT_STAGES=1 is the older spelling of 0)
  int sweep_ds = 1;       // QTOS_SWEEP_DS       0: k_step forms ds = Ji dx itself
  int spec_jac = 1;       // QTOS_SPEC_JAC       0: the first trial point of a line search is evaluated without its Jacobian
  int spec_pattern = 1;   // QTOS_SPEC_PATTERN   0: qtos_plan_submit queues the first iteration only (no launch pattern)
  int kron = 0;           // QTOS_KRON           experiment builds only: Kronecker assembly of the range-of-motion blocks
  int order = -1;         // QTOS_ORDER          time keys of the elimination order: -1 unset (reduced base: rules 2 and 1 are built, the smaller front
                          //                     is kept), 0 | 1 | 2 force rule 0 (rounds 1 - 5) / rule 1 (late force nodes) / rule 2 (early
                          //                     coefficients; model.hpp HostModel::order_rule)
  int place = 0;          // QTOS_PLACE          slot placement rule of the analysis (0 = the measured best; 1 .. 4 variants)
  int debug = 0;          // QTOS_DEBUG_SYMBOLIC (1), QTOS_DEBUG_SYMBOLIC2 (2): the analysis talks on stderr
  int debug_kron = 0;     // QTOS_DEBUG_KRON     qtos_analyze reports the Kronecker structure
  std::string dump_first; // QTOS_DUMP_FIRST     file that receives the envelope of the ordered matrix

  static QtosEnv parse() {
    QtosEnv e;
    auto num = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
    e.kkt = num("QTOS_KKT", 0);
    e.lanes = std::max(1, std::min(4, num("QTOS_LANES", 1)));
    if (getenv("QTOS_SHORT_STAGES")) e.short_stages = num("QTOS_SHORT_STAGES", 1) != 0;
    if (num("QTOS_NO_SHORT_STAGES", 0) != 0) e.short_stages = 0;
    e.sweep_ds = num("QTOS_SWEEP_DS", 1) != 0;
    e.spec_jac = num("QTOS_SPEC_JAC", 1) != 0;
    e.spec_pattern = num("QTOS_SPEC_PATTERN", 1) != 0;
    e.kron = num("QTOS_KRON", 0) != 0;
    e.place = num("QTOS_PLACE", 0);
    if (getenv("QTOS_ORDER")) e.order = std::max(0, std::min(2, num("QTOS_ORDER", 0)));
    e.debug = getenv("QTOS_DEBUG_SYMBOLIC2") ? 2 : (getenv("QTOS_DEBUG_SYMBOLIC") ? 1 : 0);
    e.debug_kron = getenv("QTOS_DEBUG_KRON") != nullptr;
    if (const char *v = getenv("QTOS_DUMP_FIRST")) e.dump_first = v;
    return e;
  }
  std::string describe() const {
    char b[320];
    snprintf(b, sizeof(b), "QTOS_KKT=%d QTOS_LANES=%d QTOS_SHORT_STAGES=%s QTOS_SWEEP_DS=%d QTOS_SPEC_JAC=%d QTOS_SPEC_PATTERN=%d QTOS_KRON=%d QTOS_PLACE=%d QTOS_ORDER=%s QTOS_DEBUG_SYMBOLIC=%d",
             kkt, lanes, short_stages < 0 ? "unset" : (short_stages ? "1" : "0"), sweep_ds, spec_jac, spec_pattern, kron, place,
             order < 0 ? "unset" : (order == 2 ? "2" : order ? "1" : "0"), debug);
    return b;
  }
};

}  // namespace qtos
