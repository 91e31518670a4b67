// csv_writer.hpp -- the plan CSV as text: 37 columns, no header, one row per millisecond (host code, no GPU).
//
// The reference's producer is the solver's own C++ stream (build/traj.csv, fetched by `docker cp`: scripts/main.py:90-92,
// QTOS/utils.py:16,19): default precision 6, i.e. printf's "%g".  A plan is 5001 rows = 185 037 numbers; formatted one by one in
// Python that is 26 ms of the 28 ms a single plan takes from its flags to the file (solve 1.6 ms, sampling 0.2 ms: round 6,
// scratch/r6_single_plan.py), so the writer is native: a "%g" of its own for the values that decide nothing but speed (finite,
// 1e-30 <= |v| < 1e30, the scaled value not within 1e-6 of a rounding boundary -- the product with a power of ten is good to
// 2e-10 there) and snprintf for everything else, so that every byte is the one printf would print
// (tests/test_boundary.py compares with Python's "%g" on random values, boundaries and the golden plan); rows are formatted in
// parallel chunks by a few threads and written with one fwrite.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace qtos {

// snprintf's "%g" with the decimal point of the C locale whatever LC_NUMERIC the host application has set (a comma there would
// be a column separator here)
inline int format_g6_libc(double v, char *out) {
  const int k = std::snprintf(out, 32, "%g", v);
  for (int i = 0; i < k; ++i)
    if (out[i] == ',') out[i] = '.';
  return k;
}

// "%g" of v into out (no terminator); returns the number of characters
inline int format_g6(double v, char *out) {
  if (v == 0.0) {
    if (std::signbit(v)) { out[0] = '-'; out[1] = '0'; return 2; }
    out[0] = '0';
    return 1;
  }
  const double a = std::fabs(v);
  if (!(a >= 1e-30 && a < 1e30)) return format_g6_libc(v, out);   // inf, nan, the far ends of the range
  static const double P10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18,
                               1e19, 1e20, 1e21, 1e22, 1e23, 1e24, 1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31, 1e32, 1e33, 1e34, 1e35, 1e36, 1e37};
  // decimal exponent from the binary one (floor(e2 log10 2) is e or e - 1), settled by the range of the scaled value.  The
  // scaling is one multiplication or division by a power of ten: at most 1.5 ulp of a number below 1e6, i.e. 2e-10 -- the
  // values it could mis-round sit within that of a rounding boundary and go to snprintf with everything within 1e-6 of one
  int e = (int)std::floor(std::ilogb(a) * 0.30102999566398120);
  auto scaled = [&](int ex) { return 5 - ex >= 0 ? a * P10[5 - ex] : a / P10[ex - 5]; };
  double s = scaled(e);
  if (s >= 1000000.0) { ++e; s = scaled(e); }
  else if (s < 100000.0) { --e; s = scaled(e); }
  uint32_t n = (uint32_t)s;
  const double frac = s - (double)n;
  if (std::fabs(frac - 0.5) < 1e-6 || !(s >= 100000.0 && s < 1000000.0)) return format_g6_libc(v, out);   // a tie to the eye: printf decides
  if (frac > 0.5) ++n;
  if (n == 1000000u) { n = 100000u; ++e; }
  char dg[6];
  for (int i = 5; i >= 0; --i) { dg[i] = (char)('0' + n % 10); n /= 10; }
  int nd = 6;
  while (nd > 1 && dg[nd - 1] == '0') --nd;        // %g strips trailing zeros
  char *p = out;
  if (v < 0) *p++ = '-';
  if (e < -4 || e >= 6) {                          // d.ddddde+XX
    *p++ = dg[0];
    if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = dg[i]; }
    *p++ = 'e';
    int ee = e;
    if (ee < 0) { *p++ = '-'; ee = -ee; } else *p++ = '+';
    *p++ = (char)('0' + ee / 10);
    *p++ = (char)('0' + ee % 10);
  } else if (e >= 0) {                             // ddd.ddd
    for (int i = 0; i <= e; ++i) *p++ = i < nd ? dg[i] : '0';
    if (nd > e + 1) { *p++ = '.'; for (int i = e + 1; i < nd; ++i) *p++ = dg[i]; }
  } else {                                         // 0.000ddd
    *p++ = '0'; *p++ = '.';
    for (int i = -1; i > e; --i) *p++ = '0';
    for (int i = 0; i < nd; ++i) *p++ = dg[i];
  }
  return (int)(p - out);
}

// rows [r0, r1) of a row-major n x cols table as CSV text
inline void format_rows(const double *rows, int cols, int r0, int r1, std::string &out) {
  out.clear();
  out.reserve((size_t)(r1 - r0) * cols * 11 + 64);   // (a plan's rows average ten characters per value; append grows it if not)
  std::vector<char> line((size_t)cols * 26);          // (a value is at most 24 characters + its separator)
  for (int r = r0; r < r1; ++r) {
    const double *x = rows + (size_t)r * cols;
    char *p = line.data();
    for (int c = 0; c < cols; ++c) {
      p += format_g6(x[c], p);
      *p++ = c + 1 < cols ? ',' : '\n';
    }
    out.append(line.data(), (size_t)(p - line.data()));
  }
}

// the whole file; n_threads <= 0: as many as the rows are worth (one per 512 rows, at most 8)
inline int write_csv_file(const char *path, const double *rows, int n_rows, int cols, int n_threads) {
  if (!path || !rows || n_rows < 0 || cols < 1) return -1;
  int nt = n_threads > 0 ? n_threads : std::min(8, std::max(1, n_rows / 512));
  nt = std::max(1, std::min(nt, std::max(1, n_rows)));
  std::vector<std::string> parts((size_t)nt);
  if (nt == 1) format_rows(rows, cols, 0, n_rows, parts[0]);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
      const int r0 = (int)((long long)n_rows * t / nt), r1 = (int)((long long)n_rows * (t + 1) / nt);
      th.emplace_back([&, t, r0, r1] { format_rows(rows, cols, r0, r1, parts[(size_t)t]); });
    }
    for (auto &x : th) x.join();
  }
  FILE *f = std::fopen(path, "w");
  if (!f) return -2;
  bool ok = true;
  for (const auto &s : parts) ok = ok && std::fwrite(s.data(), 1, s.size(), f) == s.size();
  ok = (std::fclose(f) == 0) && ok;
  return ok ? 0 : -3;
}

}  // namespace qtos
