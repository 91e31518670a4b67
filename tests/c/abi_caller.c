/* A C99 caller of the boundary: include/qtos_planner.h compiled by gcc as plain C, linked against libqtos_planner.so -- what a
 * maintainer's cgo / JNI / FFI stub does.  Host-only entry points (no GPU): the analysis of a transcription whose
 * QtosParams image the Python mirror (capi.params_from_config) wrote to a file -- sizeof and field offsets of the struct must
 * agree between the header and the ctypes mirror for the dimensions to come out right --, the build flags, the CSV writer.
 * usage: abi_caller <params.bin> <out.csv> */
#include "qtos_planner.h"

#include <stdio.h>
#include <string.h>

int main(int argc, char **argv) {
  QtosParams p;
  QtosDims d;
  int act[1024];
  double rows[3 * QTOS_CSV_COLS];
  FILE *f;
  size_t n;
  int rc, i, extra;
  if (argc != 3) return 2;
  f = fopen(argv[1], "rb");
  if (!f) return 3;
  n = fread(&p, 1, sizeof p, f);
  extra = fgetc(f) != EOF;          /* the image must be exactly one struct long */
  fclose(f);
  if (n != sizeof p || extra) { printf("params image %lu bytes (+%d), struct %lu\n", (unsigned long)n, extra, (unsigned long)sizeof p); return 4; }
  memset(&d, 0, sizeof d);
  rc = qtos_analyze(&p, &d, act, 1024);
  printf("analyze rc=%d sizeof_params=%lu sizeof_dims=%lu n_vars=%d n_cons=%d n_free=%d n_eq=%d n_ineq=%d n_unknowns=%d n_stages=%d pivots=%d front=%d order_rule=%d max_active=%d rows=%d\n",
         rc, (unsigned long)sizeof p, (unsigned long)sizeof d, d.n_vars, d.n_cons, d.n_free, d.n_eq, d.n_ineq, d.n_unknowns, d.n_stages, d.pivots, d.front,
         d.order_rule, d.max_active, d.n_rows_csv);
  printf("build_flags=%d\n", qtos_build_flags());
  for (i = 0; i < 3 * QTOS_CSV_COLS; ++i) rows[i] = 0.0;
  rows[0] = 3.756; rows[3] = 0.24; rows[QTOS_CSV_COLS] = 3.757; rows[QTOS_CSV_COLS + 1] = 6.9309e-07; rows[2 * QTOS_CSV_COLS] = 3.758;
  rc = qtos_write_csv(argv[2], rows, 3, 1);
  printf("write_csv rc=%d bad_path rc=%d null rc=%d\n", rc, qtos_write_csv("/no/such/dir/x.csv", rows, 3, 1), qtos_write_csv(argv[2], 0, 3, 1));
  return rc;
}
