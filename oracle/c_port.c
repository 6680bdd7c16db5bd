/* ORACLE (C port) — TEST / BASELINE INFRASTRUCTURE ONLY, never on the product path.
 *
 * Restates, as plain C, the loops the reference's C linker generates and runs for the
 * benchmark graphs, so that bench.py's `cpu_baseline` ("kind": "port") times the same work
 * the reference does on the host (the Python reference itself cannot travel to the GPU box).
 * Compiled with the reference's own flags (-O3 -fno-math-errno -march=native,
 * link/c/cmodule.py:2047 GCC_compiler.compile_args); single-threaded like the reference
 * (openmp defaults to False, configdefaults.py:1037).
 *
 * BASELINE config 2, FAST_RUN graph:
 *   Sum{acc_dtype=float64}(Elemwise{Composite{exp(((i0*sqr(i1-i2))/i3))}}(-0.5, x, mu, sqr(sigma)))
 *   - the Composite runs the contiguous fast path of Elemwise._c_all (tensor/elemwise.py:1070-1158):
 *     a freshly allocated N-element output (make_alloc, elemwise_cgen.py:174) and one `for i<n` loop
 *     whose body is Composite.c_code_template (scalar/basic.py:4250): one temporary per scalar op;
 *   - the Sum is a second, sequential pass over that intermediate (CAReduce._c_all :1522,
 *     make_loop_careduce elemwise_cgen.py:502) accumulating in double.
 * Pinned by tests/test_oracle_cport.py against the reference's golden output for this graph.
 */
#include <math.h>
#include <stdlib.h>

double cport_cfg2_eval(const double* x, long n, double mu, double sigma) {
  /* node: Elemwise{square}(sigma) on a (1,1) operand */
  const double sig2 = sigma * sigma;
  const double c = -0.5;
  /* node: Elemwise{Composite}: allocate output, contiguous loop */
  double* tmp = (double*)malloc((size_t)n * sizeof(double));
  if (!tmp) return NAN;
  for (long i = 0; i < n; ++i) {
    const double t0 = x[i] - mu;      /* sub  */
    const double t1 = t0 * t0;        /* sqr  */
    const double t2 = c * t1;         /* mul  */
    const double t3 = t2 / sig2;      /* true_div */
    tmp[i] = exp(t3);                 /* exp  */
  }
  /* node: Sum{acc_dtype=float64}: sequential accumulation over the intermediate */
  double acc = 0.0;
  for (long i = 0; i < n; ++i) acc += tmp[i];
  free(tmp);
  return acc;
}

/* The same graph as the reference runs it under AESARA_FLAGS=openmp=True: only the Elemwise
 * loop carries `#pragma omp parallel for` (tensor/elemwise.py:1108-1123, above
 * config.openmp_elemwise_minsize); the CAReduce loop has no OpenMP form and stays sequential. */
double cport_cfg2_eval_omp(const double* x, long n, double mu, double sigma, int threads) {
  const double sig2 = sigma * sigma;
  const double c = -0.5;
  double* tmp = (double*)malloc((size_t)n * sizeof(double));
  if (!tmp) return NAN;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < n; ++i) {
    const double t0 = x[i] - mu;
    const double t1 = t0 * t0;
    const double t2 = c * t1;
    const double t3 = t2 / sig2;
    tmp[i] = exp(t3);
  }
  double acc = 0.0;
  for (long i = 0; i < n; ++i) acc += tmp[i];
  free(tmp);
  return acc;
}

/* BASELINE config 1b: Elemwise{add,no_inplace} on two C-contiguous matrices (fresh output) */
void cport_cfg1b_add(const double* x, const double* y, double* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = x[i] + y[i];
}
