// Host-side check of poisson_cnn_amd/csrc/fft_regs.h (the in-register FFTs of the spectral transform kernels) against a double-precision DFT.
// Built and run by tests/test_fft_regs.py with g++ - no GPU needed: the header is the same source the HIP kernels include.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../poisson_cnn_amd/csrc/fft_regs.h"
using namespace pcnn_fft;
typedef std::complex<double> cd;

static double frand() { return 2.0 * rand() / RAND_MAX - 1.0; }

template <int N> static double check_cfft() {
  float re[N], im[N], re0[N], im0[N];
  for (int i = 0; i < N; ++i) { re0[i] = re[i] = (float)frand(); im0[i] = im[i] = (float)frand(); }
  cfft_dif<N, -1>(re, im);
  double err = 0, nrm = 0;
  std::vector<cd> X(N);
  for (int k = 0; k < N; ++k) {
    cd s = 0;
    for (int n = 0; n < N; ++n) s += cd(re0[n], im0[n]) * std::polar(1.0, -2 * M_PI * k * n / N);
    X[k] = s;
    const int p = bitrev(k, N);
    err += std::norm(cd(re[p], im[p]) - s); nrm += std::norm(s);
  }
  double e1 = std::sqrt(err / nrm);
  // undo: back to N x
  cfft_undo_dif<N>(re, im);
  err = nrm = 0;
  for (int n = 0; n < N; ++n) { err += std::norm(cd(re[n], im[n]) / (double)N - cd(re0[n], im0[n])); nrm += std::norm(cd(re0[n], im0[n])); }
  double e2 = std::sqrt(err / nrm);
  // unnormalised inverse by DIF with the opposite sign, natural input
  float r2[N], i2[N];
  for (int k = 0; k < N; ++k) { r2[k] = (float)X[k].real(); i2[k] = (float)X[k].imag(); }
  cfft_dif<N, +1>(r2, i2);
  err = nrm = 0;
  for (int n = 0; n < N; ++n) { const int p = bitrev(n, N); err += std::norm(cd(r2[p], i2[p]) / (double)N - cd(re0[n], im0[n])); nrm += std::norm(cd(re0[n], im0[n])); }
  double e3 = std::sqrt(err / nrm);
  return std::fmax(e1, std::fmax(e2, e3));
}

template <int N> static double check_rfft() {
  float x[N], x0[N];
  for (int i = 0; i < N; ++i) x0[i] = x[i] = (float)frand();
  rfft_fwd<N>(x);
  double err = 0, nrm = 0;
  for (int f = 0; f <= N / 2; ++f) {
    cd s = 0;
    for (int n = 0; n < N; ++n) s += (double)x0[n] * std::polar(1.0, -2 * M_PI * f * n / N);
    bool neg;
    const double gr = x[rfft_pos(N, f, false, neg)];
    double gi = 0;
    if (f != 0 && f != N / 2) { const int p = rfft_pos(N, f, true, neg); gi = neg ? -x[p] : x[p]; }
    err += std::norm(cd(gr, gi) - s); nrm += std::norm(s);
  }
  // the N positions are a permutation
  std::vector<int> seen(N, 0);
  for (int f = 0; f <= N / 2; ++f) {
    bool neg;
    seen[rfft_pos(N, f, false, neg)]++;
    if (f != 0 && f != N / 2) seen[rfft_pos(N, f, true, neg)]++;
  }
  for (int i = 0; i < N; ++i) if (seen[i] != 1) { printf("rfft_pos<%d> is not a permutation at %d\n", N, i); exit(2); }
  const double e1 = std::sqrt(err / nrm);
  rfft_inv<N>(x);
  err = nrm = 0;
  for (int n = 0; n < N; ++n) { err += (x[n] / (double)N - x0[n]) * (x[n] / (double)N - x0[n]); nrm += (double)x0[n] * x0[n]; }
  return std::fmax(e1, std::sqrt(err / nrm));
}

int main() {
  srand(12345);
  double worst = 0;
  for (int rep = 0; rep < 200; ++rep) {
    worst = std::fmax(worst, check_cfft<2>()); worst = std::fmax(worst, check_cfft<4>()); worst = std::fmax(worst, check_cfft<8>());
    worst = std::fmax(worst, check_cfft<16>()); worst = std::fmax(worst, check_cfft<32>()); worst = std::fmax(worst, check_cfft<64>());
    worst = std::fmax(worst, check_rfft<2>()); worst = std::fmax(worst, check_rfft<4>()); worst = std::fmax(worst, check_rfft<8>());
    worst = std::fmax(worst, check_rfft<16>()); worst = std::fmax(worst, check_rfft<32>()); worst = std::fmax(worst, check_rfft<64>());
  }
  printf("worst rel-L2 error %.3g\n", worst);
  return worst < 5e-7 ? 0 : 1;
}
