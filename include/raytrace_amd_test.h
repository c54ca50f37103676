/*
 * raytrace_amd_test.h — entry points of libraytrace_amd.so that exist for its tests only: they evaluate, on
 * caller-given values, arithmetic the kernels use internally and no caller of the renderer needs.  Not part of the
 * drop-in boundary (include/raytrace_amd.h).
 */
#ifndef RAYTRACE_AMD_TEST_H
#define RAYTRACE_AMD_TEST_H

#include "raytrace_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Replaces: nothing callable in the reference — GLSL's sqrt() inside hit_sphere, normalize() and refract()
 * (testFragment.fs:151, 246-287), which this library takes to be the correctly rounded binary32 square root.  The
 * kernels compute it with a shortened instruction sequence wherever a whole wavefront holds ordinary arguments;
 * these two entry points let a test hold that sequence to the full one.  rt_eval_sqrt: y[i] = that square root of
 * x[i], n values, host pointers.  rt_check_sqrt: over the `count` binary32 bit patterns from first_bits on (all
 * 2^32 in one call if asked), *mismatches = how many differ in any bit from the compiler's correctly rounded sqrtf
 * (two NaNs are equal) and *example_bits (nullable) = one of them.
 */
RT_API int rt_eval_sqrt(rt_ctx* ctx, const float* x, size_t n, float* y);
RT_API int rt_check_sqrt(rt_ctx* ctx, uint64_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* example_bits);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACE_AMD_TEST_H */
