// CPU model of the wave-parallel exact evaluation of the sequential chain acc = fl(acc + d_i), first acc >= p.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
typedef unsigned long long u64;
static inline u64 bits(double x) { u64 b; memcpy(&b, &x, 8); return b; }
static inline double fromb(u64 b) { double x; memcpy(&x, &b, 8); return x; }

static int seq_pick(const double *d, int n, double p, double *acc_out) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) { acc = acc + d[i]; if (acc >= p) { *acc_out = acc; return i; } }
  *acc_out = acc; return -1;
}

// one element's map N -> N + c[N & 1] in units of u = 2^(e - 52), e = exponent of acc; D >= 2^53 marks "leaves the binade"
static void elem_map(double d, int e, u64 *c0, u64 *c1) {
  const u64 BIG = 1ull << 53;
  if (d == 0.0) { *c0 = *c1 = 0; return; }
  u64 b = bits(d);
  int ed = (int)((b >> 52) & 0x7FF);
  if (ed == 0 || ed == 0x7FF || (b >> 63)) { *c0 = *c1 = BIG; return; }   // subnormal / inf / nan / negative: sequential
  u64 md = (b & ((1ull << 52) - 1)) | (1ull << 52);
  int shift = e - (ed - 1023);
  if (shift <= 0) { *c0 = *c1 = BIG; return; }
  if (shift >= 64) { *c0 = *c1 = 0; return; }
  u64 D = md >> shift, rem = md & ((1ull << shift) - 1ull), half = 1ull << (shift - 1);
  if (rem > half) { *c0 = *c1 = D + 1; }
  else if (rem < half) { *c0 = *c1 = D; }
  else { *c0 = D + (D & 1ull); *c1 = D + ((D + 1ull) & 1ull); }
}

static int fast_pick(const double *d, int n, double p, double *acc_out, long *n_seq, long *n_par) {
  double acc = 0.0;
  for (int base = 0; base < n; base += 64) {
    int cnt = n - base < 64 ? n - base : 64;
    int start = 0;
    while (start < cnt) {
      u64 ab = bits(acc);
      int ea = (int)((ab >> 52) & 0x7FF);
      int f = -1;                                   // first lane >= start that leaves the binade (or must go sequentially)
      u64 N[64];
      if (ea == 0 || ea == 0x7FF || (ab >> 63)) f = start;           // acc zero / subnormal / not finite: one sequential step
      else {
        int e = ea - 1023;
        u64 N0 = (ab & ((1ull << 52) - 1)) | (1ull << 52);
        u64 c0[64], c1[64];
        for (int l = 0; l < 64; ++l) { if (l >= start && l < cnt) elem_map(d[base + l], e, &c0[l], &c1[l]); else c0[l] = c1[l] = 0; }
        // inclusive scan of the maps (Hillis-Steele, as the wave would do it)
        for (int off = 1; off < 64; off <<= 1) {
          u64 n0[64], n1[64];
          for (int l = 0; l < 64; ++l) {
            if (l >= off) { u64 f0 = c0[l - off], f1 = c1[l - off], g0 = c0[l], g1 = c1[l];
              n0[l] = f0 + ((f0 & 1ull) ? g1 : g0);
              n1[l] = f1 + (((1ull + f1) & 1ull) ? g1 : g0);
            } else { n0[l] = c0[l]; n1[l] = c1[l]; }
          }
          memcpy(c0, n0, sizeof n0); memcpy(c1, n1, sizeof n1);
        }
        for (int l = 0; l < 64; ++l) N[l] = N0 + ((N0 & 1ull) ? c1[l] : c0[l]);
        for (int l = start; l < cnt; ++l) if (N[l] >= (1ull << 53)) { f = l; break; }
        // hits among the lanes before the crossing
        int lim = f < 0 ? cnt : f;
        for (int l = start; l < lim; ++l) {
          double a = ldexp((double)N[l], e - 52);
          if (a >= p) { *acc_out = a; return base + l; }
        }
        ++*n_par;
        if (f < 0) { acc = ldexp((double)N[cnt - 1], e - 52); break; }
        if (f > start) acc = ldexp((double)N[f - 1], e - 52);
      }
      acc = acc + d[base + f]; ++*n_seq;            // the element that leaves the binade: a real addition
      if (acc >= p) { *acc_out = acc; return base + f; }
      start = f + 1;
    }
  }
  *acc_out = acc; return -1;
}

static double urand(void) { return (double)rand() / ((double)RAND_MAX + 1.0); }
int main(int argc, char **argv) {
  long bad = 0, tests = 0, nseq = 0, npar = 0;
  srand(12345);
  for (int t = 0; t < 20000; ++t) {
    int n = 1 + rand() % 3000;
    if (t % 50 == 0) n = 20000 + rand() % 200000;
    double *w = malloc(sizeof(double) * n), *d = malloc(sizeof(double) * n);
    int kind = t % 7;
    double S = 0.0;
    for (int i = 0; i < n; ++i) {
      float x;
      switch (kind) {
        case 0: x = 1.0f; break;                                     // unweighted
        case 1: x = (float)(1 + rand() % 4) * 0.25f; break;          // p/q variants of unit weights
        case 2: x = (float)urand(); break;
        case 3: x = (float)exp(20.0 * urand() - 10.0); break;        // wide dynamic range
        case 4: x = (i % 97 == 0) ? 0.0f : (float)urand(); break;    // zeros
        case 5: x = (float)ldexp(1.0, rand() % 12 - 6); break;       // powers of two: ties galore
        default: x = (float)((rand() % 1000) + 1); break;
      }
      w[i] = (double)x; S += w[i];
    }
    for (int i = 0; i < n; ++i) d[i] = w[i] / S;
    for (int r = 0; r < 6; ++r) {
      double p = (double)(rand() % (1 << 24)) * 0x1p-24;
      if (r == 5) p = 2.0;                                            // never reached: whole chain
      double a1, a2; int k1 = seq_pick(d, n, p, &a1), k2 = fast_pick(d, n, p, &a2, &nseq, &npar);
      ++tests;
      if (k1 != k2 || bits(a1) != bits(a2)) { if (bad < 10) printf("MISMATCH kind %d n %d p %.17g: seq %d %.17g fast %d %.17g\n", kind, n, p, k1, a1, k2, a2); ++bad; }
    }
    free(w); free(d);
  }
  printf("tests %ld bad %ld; sequential single steps %ld, parallel rounds %ld\n", tests, bad, nseq, npar);
  return bad != 0;
}
