/*
 * conv3p CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see conv3p_oracle.h).
 *
 * Plain C restatement of /root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp
 * (CPU Conv3p / Conv3pGrad).  The body lives in conv3p_oracle_body.inc and is
 * instantiated for float and double, the two dtypes the reference registers
 * (.cpp:515-516, :726-727).
 *
 * Build exactly like the reference CPU object as far as floating point goes
 * (tf_conv3p_compile.sh:32: -O3, no -march, no -ffast-math): x86-64 baseline has
 * no FMA, and -ffp-contract=off keeps it that way on any host.
 *
 * PARITY STATUS: pinned.  Neighbour search against the reference Grid template and the
 * accumulation loops bit-for-bit against the reference's own loop text, both compiled in
 * place into oracle/_ref (oracle/Makefile, tests/test_oracle.py).
 */
#include "conv3p_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_thread_num(void) { return 0; }
static int omp_get_max_threads(void) { return 1; }
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SFX(name) CAT(name, _f32)
#include "conv3p_oracle_body.inc"
#undef REAL
#undef SFX

#define REAL double
#define SFX(name) CAT(name, _f64)
#include "conv3p_oracle_body.inc"
#undef REAL
#undef SFX

int conv3p_oracle_max_threads(void) { return omp_get_max_threads(); }
