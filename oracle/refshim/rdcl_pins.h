/* TEST INFRASTRUCTURE ONLY (oracle/).
 * Forced into the reference's OpenCL programs when they are built by a REAL OpenCL device's compiler (-Wf,-include,<this file> through the AMD runtime's
 * AMD_OCL_BUILD_OPTIONS_APPEND; tools/gpu_probe_ocl2.sh, tools/ref_ops_on_opencl.py): the three builtins whose accuracy OpenCL leaves to the device (rsqrt 2 ulp,
 * hypot 4 ulp, distance ~2.5 ulp) take the definitions our serial stand-in uses (rdcl_builtins.c), so that what is left of a difference to the goldens is not the device's choice of
 * these.  The reference's sources stay untouched; every other builtin is the device's own. */
#pragma OPENCL EXTENSION cl_khr_fp64 : enable
inline float __attribute__((overloadable)) rdpin_len(float2 d) { return sqrt(d.x * d.x + d.y * d.y); }
inline float __attribute__((overloadable)) rdpin_len(float3 d) { return sqrt(d.x * d.x + d.y * d.y + d.z * d.z); }
#define distance(a, b) rdpin_len((a) - (b))
#define rsqrt(x) (1.0f / sqrt(x))
#define hypot(a, b) ((float)sqrt((double)(a) * (double)(a) + (double)(b) * (double)(b)))
