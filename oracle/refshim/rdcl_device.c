/*
 * TEST INFRASTRUCTURE ONLY (oracle/). Never linked into the product library.
 *
 * A minimal "serial CPU device" behind the OpenCL 1.2 entry points that the reference's
 * host C (oclhelper.c, oclimgutil.c, oclpolyline.c, oclrect.c) and its apps import
 * (25 + 3 cl* symbols).  This image has the OpenCL headers and ICD loader
 * but NO OpenCL device, so the reference cannot run through a real OpenCL runtime here.
 *
 *  - clCreateProgramWithSource / clBuildProgram: the three reference programs are
 *    recognised by a kernel name in their source text and bound to the x86-64 shared
 *    object that oracle/Makefile compiled from the very same .cl file
 *    (oracle/_ref/{oclimgutil,oclpolyline,oclrect}_k.so).
 *  - clEnqueueNDRangeKernel: executes every work-item SERIALLY, dimension 0 fastest
 *    ("raster order", SURVEY.md 7.3 canonical semantics), by calling the kernel's
 *    C-ABI function with the clSetKernelArg slots marshalled per the SysV ABI.
 *  - buffers are host allocations, zero-filled when created without a host pointer.
 *  - a trace facility records every launch and can snapshot any buffer argument right
 *    after a chosen launch, so tests can look at the reference's intermediate planes.
 */
#define _GNU_SOURCE
#define CL_TARGET_OPENCL_VERSION 120
#define CL_USE_DEPRECATED_OPENCL_1_2_APIS
#include <CL/cl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXARGS 16
#define NPROG 3

struct _cl_platform_id { int dummy; };
struct _cl_device_id { int dummy; };
struct _cl_context { int refs; };
struct _cl_command_queue { int refs; };
struct _cl_mem { uint32_t magic; void *data; size_t size; struct _cl_mem *next, *prev; };
struct _cl_program { int which; void *dl; size_t *gid; void (*defer_min)(int); };
struct _cl_kernel {
  struct _cl_program prog;
  char name[64];
  void *fn;
  int nargs;
  size_t argsize[MAXARGS];
  uint64_t argval[MAXARGS];
  uint32_t floatmask;
};
struct _cl_event { int refs; };

#define MEM_MAGIC 0x52444d45u

static struct _cl_platform_id the_platform;
static struct _cl_device_id the_device;
static struct _cl_mem *live_mems = NULL;

static const char *prog_tag[NPROG] = { "imgutil", "polyline", "rect" };
static const char *prog_so[NPROG] = { "oclimgutil_k.so", "oclpolyline_k.so", "oclrect_k.so" };
static void *prog_dl[NPROG];

/* ---------------------------------------------------------------- trace */

typedef struct { char name[96]; size_t gws[2]; int dim; uint64_t hash[16]; size_t bytes[16]; } rdcl_launch_t;
typedef struct { char name[96]; int occurrence, argidx, ordinal; void *data; size_t size; int done; } rdcl_snap_t;

static rdcl_launch_t *trace_buf = NULL;
static int trace_n = 0, trace_cap = 0, trace_on = 1;
static rdcl_snap_t snaps[256];
static int nsnaps = 0;

void rdcl_trace_reset(void) { trace_n = 0; }
void rdcl_trace_enable(int on) { trace_on = on; }
int rdcl_trace_count(void) { return trace_n; }
const char *rdcl_trace_name(int i) { return (i >= 0 && i < trace_n) ? trace_buf[i].name : ""; }
size_t rdcl_trace_gws(int i, int d) { return (i >= 0 && i < trace_n && d >= 0 && d < 2) ? trace_buf[i].gws[d] : 0; }
/* rdcl_hash_all(1): fingerprint and size of buffer argument `arg` of launch i right after it ran (size 0: not a buffer) */
uint64_t rdcl_trace_hash(int i, int arg) { return (i >= 0 && i < trace_n && arg >= 0 && arg < 16) ? trace_buf[i].hash[arg] : 0; }
size_t rdcl_trace_bytes(int i, int arg) { return (i >= 0 && i < trace_n && arg >= 0 && arg < 16) ? trace_buf[i].bytes[arg] : 0; }

static size_t snap_limit = 0;      /* snapshots keep at most so many bytes of a buffer (0: all of it) */
void rdcl_snapshot_limit(size_t bytes) { snap_limit = bytes; }
void rdcl_zero_fill(int on) { (void)on; }      /* (the observer's switch, rdcl_observe.c: this device's buffers always start as zeros) */

void rdcl_snapshot_clear(void) {
  for (int i = 0; i < nsnaps; i++) free(snaps[i].data);
  nsnaps = 0;
}

/* Ask for a copy of buffer argument `argidx` of the `occurrence`-th (0-based, counted since
 * the last rdcl_trace_reset) launch of kernel "<prog>:<kernel>", taken right after it ran.
 * occurrence < 0 means "the launch with global ordinal -occurrence-1". Returns a handle. */
int rdcl_snapshot_request(const char *name, int occurrence, int argidx) {
  if (nsnaps >= 256) return -1;
  rdcl_snap_t *s = &snaps[nsnaps];
  memset(s, 0, sizeof(*s));
  snprintf(s->name, sizeof(s->name), "%s", name);
  s->occurrence = occurrence;
  s->argidx = argidx;
  s->ordinal = -1;
  return nsnaps++;
}

int rdcl_snapshot_fetch(int h, void **data, size_t *size, int *ordinal) {
  if (h < 0 || h >= nsnaps || !snaps[h].done) return -1;
  *data = snaps[h].data;
  *size = snaps[h].size;
  if (ordinal) *ordinal = snaps[h].ordinal;
  return 0;
}

/* ---------------------------------------------------------------- helpers */

static void self_dir(char *out, size_t n) {
  Dl_info info;
  out[0] = '\0';
  if (dladdr((void *)&self_dir, &info) && info.dli_fname) {
    snprintf(out, n, "%s", info.dli_fname);
    char *s = strrchr(out, '/');
    if (s) s[1] = '\0'; else snprintf(out, n, "./");
  }
}

static struct _cl_mem *find_mem(uint64_t v) {
  for (struct _cl_mem *m = live_mems; m; m = m->next) if ((uint64_t)(uintptr_t)m == v) return m;
  return NULL;
}

static uint32_t float_arg_mask(const char *kname) {
  /* 4-byte clSetKernelArg slots that are `float` parameters in the .cl text */
  if (!strcmp(kname, "cast_i_f")) return 1u << 2;
  if (!strcmp(kname, "threshold_f_f")) return (1u << 2) | (1u << 3) | (1u << 4);
  if (!strcmp(kname, "threshold_f")) return (1u << 1) | (1u << 2) | (1u << 3);
  if (!strcmp(kname, "convert_bgr_lumaf")) return 1u << 2;
  if (!strcmp(kname, "convert_bgr_luminancef")) return 1u << 2;
  if (!strcmp(kname, "mkpl_pass2")) return 1u << 9;
  if (!strcmp(kname, "mkpl_pass4")) return 1u << 4;
  return 0;
}

/* ---------------------------------------------------------------- platform / device */

cl_int clGetPlatformIDs(cl_uint n, cl_platform_id *p, cl_uint *np) {
  if (p && n > 0) p[0] = &the_platform;
  if (np) *np = 1;
  return CL_SUCCESS;
}

cl_int clGetDeviceIDs(cl_platform_id p, cl_device_type t, cl_uint n, cl_device_id *d, cl_uint *nd) {
  (void)p; (void)t;
  if (d && n > 0) d[0] = &the_device;
  if (nd) *nd = 1;
  return CL_SUCCESS;
}

cl_int clGetDeviceInfo(cl_device_id d, cl_device_info what, size_t sz, void *val, size_t *ret) {
  (void)d;
  const char *s = "";
  switch (what) {
  case CL_DEVICE_NAME: s = "rectdetect oracle serial CPU device"; break;
  case CL_DEVICE_VERSION: s = "OpenCL 1.2 serial-shim"; break;
  case CL_DEVICE_EXTENSIONS: s = ""; break; /* no int64 atomics: kernels use the 2x32-bit path */
  default: return CL_INVALID_VALUE;
  }
  size_t need = strlen(s) + 1;
  if (ret) *ret = need;
  if (val) {
    if (sz < need) return CL_INVALID_VALUE;
    memcpy(val, s, need);
  }
  return CL_SUCCESS;
}

cl_context clCreateContext(const cl_context_properties *props, cl_uint n, const cl_device_id *devs,
                           void (CL_CALLBACK *cb)(const char *, const void *, size_t, void *), void *ud, cl_int *err) {
  (void)props; (void)n; (void)devs; (void)cb; (void)ud;
  cl_context c = (cl_context)calloc(1, sizeof(*c));
  c->refs = 1;
  if (err) *err = CL_SUCCESS;
  return c;
}

cl_int clReleaseContext(cl_context c) { if (c && --c->refs == 0) free(c); return CL_SUCCESS; }

cl_command_queue clCreateCommandQueue(cl_context c, cl_device_id d, cl_command_queue_properties p, cl_int *err) {
  (void)c; (void)d; (void)p;
  cl_command_queue q = (cl_command_queue)calloc(1, sizeof(*q));
  q->refs = 1;
  if (err) *err = CL_SUCCESS;
  return q;
}

cl_int clReleaseCommandQueue(cl_command_queue q) { if (q && --q->refs == 0) free(q); return CL_SUCCESS; }
cl_int clFinish(cl_command_queue q) { (void)q; return CL_SUCCESS; }
cl_int clFlush(cl_command_queue q) { (void)q; return CL_SUCCESS; }

/* ---------------------------------------------------------------- programs / kernels */

cl_program clCreateProgramWithSource(cl_context c, cl_uint count, const char **strings, const size_t *lengths, cl_int *err) {
  (void)c; (void)lengths;
  int which = -1;
  for (cl_uint i = 0; i < count && which < 0; i++) {
    if (strstr(strings[i], "labelMergeMain")) which = 2;
    else if (strstr(strings[i], "mkpl_pass0a")) which = 1;
    else if (strstr(strings[i], "iirblur_f_f_pass0a")) which = 0;
  }
  if (which < 0) { if (err) *err = CL_INVALID_VALUE; return NULL; }
  cl_program p = (cl_program)calloc(1, sizeof(*p));
  p->which = which;
  if (err) *err = CL_SUCCESS;
  return p;
}

cl_int clBuildProgram(cl_program p, cl_uint n, const cl_device_id *d, const char *opts,
                      void (CL_CALLBACK *cb)(cl_program, void *), void *ud) {
  (void)n; (void)d; (void)opts; (void)cb; (void)ud;
  if (!prog_dl[p->which]) {
    char path[4096];
    const char *vd = getenv("RDCL_KERNEL_DIR");      /* kernel objects of a builtin variant (oracle/Makefile: ref_variants), e.g. <this directory>/var_fma */
    if (vd && *vd && strlen(vd) < sizeof(path) - 80) { strcpy(path, vd); strcat(path, "/"); }
    else self_dir(path, sizeof(path) - 64);
    strcat(path, prog_so[p->which]);
    prog_dl[p->which] = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!prog_dl[p->which]) {
      fprintf(stderr, "rdcl: cannot load %s: %s\n", path, dlerror());
      return CL_BUILD_PROGRAM_FAILURE;
    }
  }
  p->dl = prog_dl[p->which];
  p->gid = (size_t *)dlsym(p->dl, "rdcl_gid");
  p->defer_min = (void (*)(int))dlsym(p->dl, "rdcl_defer_atomic_min");
  return p->gid ? CL_SUCCESS : CL_BUILD_PROGRAM_FAILURE;
}

cl_int clGetProgramBuildInfo(cl_program p, cl_device_id d, cl_program_build_info what, size_t sz, void *val, size_t *ret) {
  (void)p; (void)d; (void)what;
  if (ret) *ret = 1;
  if (val && sz > 0) ((char *)val)[0] = '\0';
  return CL_SUCCESS;
}

cl_int clReleaseProgram(cl_program p) { free(p); return CL_SUCCESS; }

cl_kernel clCreateKernel(cl_program p, const char *name, cl_int *err) {
  void *fn = dlsym(p->dl, name);
  if (!fn) { if (err) *err = CL_INVALID_KERNEL_NAME; return NULL; }
  cl_kernel k = (cl_kernel)calloc(1, sizeof(*k));
  k->prog = *p;
  snprintf(k->name, sizeof(k->name), "%s", name);
  k->fn = fn;
  k->floatmask = float_arg_mask(name);
  if (err) *err = CL_SUCCESS;
  return k;
}

cl_int clReleaseKernel(cl_kernel k) { free(k); return CL_SUCCESS; }

cl_int clGetKernelInfo(cl_kernel k, cl_kernel_info what, size_t sz, void *val, size_t *ret) {
  if (what != CL_KERNEL_FUNCTION_NAME) return CL_INVALID_VALUE;
  size_t need = strlen(k->name) + 1;
  if (ret) *ret = need;
  if (val) { if (sz < need) return CL_INVALID_VALUE; memcpy(val, k->name, need); }
  return CL_SUCCESS;
}

cl_int clSetKernelArg(cl_kernel k, cl_uint idx, size_t size, const void *val) {
  if (idx >= MAXARGS || (size != 4 && size != 8)) return CL_INVALID_ARG_INDEX;
  k->argsize[idx] = size;
  k->argval[idx] = 0;
  memcpy(&k->argval[idx], val, size);
  if ((int)idx + 1 > k->nargs) k->nargs = idx + 1;
  return CL_SUCCESS;
}

/* ---------------------------------------------------------------- work-item order (default: raster, dimension 0 fastest)
 * OpenCL leaves the order in which work-items run to the device.  rdcl_set_order(filter, gw, gh, group_order, seed) makes the
 * launches of the kernels named in `filter` ("prog:kernel,prog:kernel,..."; NULL or "" = every kernel) run in another LEGAL
 * order, so that tests can see which of the reference's results depend on it: the NDRange is cut into
 * work-groups of gw x gh items (0 x 0: one group = the whole range), items inside a group run in raster order, and the
 * groups are visited in group_order 0 = raster, 1 = reversed raster, 2 = column-major, 3 = scrambled (a stride walk that
 * depends on `seed`).  group_order 4 = the whole range in reversed raster order (items too).  group_order 5 = CONCURRENT: no
 * work-item sees another one's update - for kernels that update memory through atomic_min only (labelMergeMain): all of them
 * read the state the launch began with and their minima take effect together at the end (rdcl_builtins.c); kernels that use
 * plain stores run in raster order as before. */
static int order_on = 0, order_gw = 0, order_gh = 0, order_go = 0, order_seed = 0;
static char order_filter[512] = "";
void rdcl_set_order(const char *filter, int gw, int gh, int group_order, int seed) {
  order_on = !(gw == 0 && gh == 0 && group_order == 0);
  order_gw = gw; order_gh = gh; order_go = group_order; order_seed = seed;
  snprintf(order_filter, sizeof(order_filter), "%s", filter ? filter : "");
}
static size_t gcd_sz(size_t a, size_t b) { while (b) { size_t t = a % b; a = b; b = t; } return a; }
static int order_applies(const struct _cl_kernel *k) {
  if (!order_on) return 0;
  if (order_filter[0] == '\0') return 1;
  char full[96];
  snprintf(full, sizeof(full), "%s:%s", prog_tag[k->prog.which], k->name);
  const size_t n = strlen(full);
  for (const char *p = order_filter; (p = strstr(p, full)) != NULL; p += n)
    if ((p == order_filter || p[-1] == ',') && (p[n] == '\0' || p[n] == ',')) return 1;
  return 0;
}


/* ---------------------------------------------------------------- interventions for tests (none is on by default)
 * rdcl_set_repeat(name, occurrence, extra): right after the `occurrence`-th launch of "<prog>:<kernel>" (counted since the last rdcl_trace_reset) the SAME launch - same
 *   arguments, same work-item order - is run up to `extra` more times; rdcl_repeat_changed(i) = 32-bit words of the kernel's buffer arguments that extra launch i changed
 *   (-1: not run; the extra launches stop after one that changed nothing - every later one would repeat it).  What it is for: is the reference's region merge settled after
 *   its 8 launches (oclrect.c:325-331), and what does the reference return when the kernel is launched until it is?
 * rdcl_skip_launches(on): launches become no-ops (the host side of the reference alone, on substituted read-backs).
 * rdcl_substitute_reads(n, data, sizes): the next n clEnqueueReadBuffer calls return the given bytes instead of the buffer's (the caller's planes in place of the three
 *   read-backs of genGPUTask, oclrect.c:371-376, so that the reference's own compiled executeCPUTask runs on them).
 * rdcl_hash_all(on): every launch of the trace also records a 64-bit hash of each of its buffer arguments, taken right after it (rdcl_trace_hash). */
#define MAXREPEAT 256
static char repeat_name[96] = "";
static int repeat_occ = -1, repeat_extra = 0, repeat_changed[MAXREPEAT];
void rdcl_set_repeat(const char *name, int occurrence, int extra) {
  snprintf(repeat_name, sizeof(repeat_name), "%s", name ? name : "");
  repeat_occ = occurrence; repeat_extra = extra > MAXREPEAT ? MAXREPEAT : extra;
  for (int i = 0; i < MAXREPEAT; i++) repeat_changed[i] = -1;
}
int rdcl_repeat_changed(int i) { return (i >= 0 && i < MAXREPEAT) ? repeat_changed[i] : -1; }
static int skip_launches = 0;
void rdcl_skip_launches(int on) { skip_launches = on; }
#define MAXSUBST 8
static const void *subst_data[MAXSUBST]; static size_t subst_size[MAXSUBST]; static int subst_n = 0, subst_at = 0;
void rdcl_substitute_reads(int n, const void **data, const size_t *sizes) {
  subst_n = n > MAXSUBST ? MAXSUBST : (n < 0 ? 0 : n); subst_at = 0;
  for (int i = 0; i < subst_n; i++) { subst_data[i] = data[i]; subst_size[i] = sizes[i]; }
}
static int hash_all = 0;
void rdcl_hash_all(int on) { hash_all = on; }
static uint64_t hash_bytes(const void *p, size_t n) {      /* FNV-1a over 64-bit words (+ the tail bytes): a fingerprint, nothing more */
  uint64_t h = 1469598103934665603ull;
  const uint64_t *w = (const uint64_t *)p;
  for (size_t i = 0; i < n / 8; i++) { h ^= w[i]; h *= 1099511628211ull; }
  const unsigned char *b = (const unsigned char *)p + (n & ~(size_t)7);
  for (size_t i = 0; i < (n & 7); i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

typedef void (*generic_fn)(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                           int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                           float, float, float, float);

/* all work-items of one launch, in the order rdcl_set_order selected for this kernel */
static void run_items(cl_kernel k, const int64_t *ia, const float *fa, size_t g0, size_t g1) {
  generic_fn fn = (generic_fn)k->fn;
  size_t *gid = k->prog.gid;
  gid[2] = 0;
#define RUN_ITEM(X, Y) do { gid[0] = (X); gid[1] = (Y); \
      fn(ia[0], ia[1], ia[2], ia[3], ia[4], ia[5], ia[6], ia[7], ia[8], ia[9], ia[10], ia[11], ia[12], ia[13], ia[14], ia[15], \
         fa[0], fa[1], fa[2], fa[3]); } while (0)
  if (!order_applies(k)) {
    for (size_t y = 0; y < g1; y++) for (size_t x = 0; x < g0; x++) RUN_ITEM(x, y);
  } else if (order_go == 5) {
    if (k->prog.defer_min) k->prog.defer_min(1);
    for (size_t y = 0; y < g1; y++) for (size_t x = 0; x < g0; x++) RUN_ITEM(x, y);
    if (k->prog.defer_min) k->prog.defer_min(0);
  } else if (order_go == 4) {
    for (size_t y = g1; y-- > 0;) for (size_t x = g0; x-- > 0;) RUN_ITEM(x, y);
  } else {
    const size_t tw = order_gw > 0 ? (size_t)order_gw : g0, th = order_gh > 0 ? (size_t)order_gh : g1;
    const size_t nx = (g0 + tw - 1) / tw, ny = (g1 + th - 1) / th, nt = nx * ny;
    size_t step = 1, start = 0;
    if (order_go == 3 && nt > 1) {
      step = (7919u + 104729u * (size_t)order_seed) % nt; if (step == 0) step = 1;
      while (gcd_sz(step, nt) != 1) step++;
      start = (nt / 3 + 31u * (size_t)order_seed) % nt;
    }
    for (size_t i = 0, t = start; i < nt; i++, t = (t + step) % nt) {
      size_t g = order_go == 3 ? t : i;
      if (order_go == 1) g = nt - 1 - i;
      const size_t gx = order_go == 2 ? g / ny : g % nx, gy = order_go == 2 ? g % ny : g / nx;
      const size_t tx = gx * tw, ty = gy * th;
      for (size_t y = ty; y < ty + th && y < g1; y++) for (size_t x = tx; x < tx + tw && x < g0; x++) RUN_ITEM(x, y);
    }
  }
#undef RUN_ITEM
}

cl_int clEnqueueNDRangeKernel(cl_command_queue q, cl_kernel k, cl_uint dim, const size_t *off, const size_t *gws,
                              const size_t *lws, cl_uint nev, const cl_event *evl, cl_event *ev) {
  (void)q; (void)off; (void)lws; (void)nev; (void)evl;
  if (dim < 1 || dim > 2) return CL_INVALID_WORK_DIMENSION;

  /* SysV x86-64: INTEGER-class parameters (ints, longs, pointers) go to rdi,rsi,rdx,rcx,r8,r9 and
   * then to consecutive 8-byte stack slots in order; float parameters go to xmm0.. in order.  The two
   * sequences are independent, so one generic signature covers every kernel (<=12 args, <=3 floats). */
  int64_t ia[16]; float fa[4]; int ni = 0, nf = 0;
  memset(ia, 0, sizeof(ia)); memset(fa, 0, sizeof(fa));
  void *argptr[MAXARGS]; size_t argbytes[MAXARGS];
  for (int i = 0; i < k->nargs; i++) {
    argptr[i] = NULL; argbytes[i] = 0;
    if (k->argsize[i] == 4 && (k->floatmask & (1u << i))) {
      if (nf >= 4) return CL_INVALID_KERNEL_ARGS;
      memcpy(&fa[nf++], &k->argval[i], 4);
    } else if (k->argsize[i] == 4) {
      int32_t v; memcpy(&v, &k->argval[i], 4);
      ia[ni++] = v;
    } else {
      struct _cl_mem *m = find_mem(k->argval[i]);
      if (m) { ia[ni++] = (int64_t)(intptr_t)m->data; argptr[i] = m->data; argbytes[i] = m->size; }
      else ia[ni++] = (int64_t)k->argval[i];
    }
  }

  const size_t g0 = gws[0], g1 = dim == 2 ? gws[1] : 1;
  if (!skip_launches) run_items(k, ia, fa, g0, g1);

  if (trace_on) {
    char full[96];
    snprintf(full, sizeof(full), "%s:%s", prog_tag[k->prog.which], k->name);
    int occ = 0;
    for (int i = 0; i < trace_n; i++) if (!strcmp(trace_buf[i].name, full)) occ++;
    if (trace_n >= trace_cap) {
      trace_cap = trace_cap ? trace_cap * 2 : 1024;
      trace_buf = (rdcl_launch_t *)realloc(trace_buf, trace_cap * sizeof(rdcl_launch_t));
    }
    snprintf(trace_buf[trace_n].name, sizeof(trace_buf[trace_n].name), "%s", full);
    trace_buf[trace_n].gws[0] = g0; trace_buf[trace_n].gws[1] = g1; trace_buf[trace_n].dim = (int)dim;
    for (int s = 0; s < nsnaps; s++) {
      rdcl_snap_t *sn = &snaps[s];
      if (sn->done) continue;
      int hit = sn->occurrence < 0 ? (trace_n == -sn->occurrence - 1) : (!strcmp(sn->name, full) && sn->occurrence == occ);
      if (!hit) continue;
      if (sn->argidx < 0 || sn->argidx >= k->nargs || !argptr[sn->argidx]) continue;
      sn->size = argbytes[sn->argidx];
      if (snap_limit && sn->size > snap_limit) sn->size = snap_limit;
      sn->data = malloc(sn->size);
      memcpy(sn->data, argptr[sn->argidx], sn->size);
      sn->ordinal = trace_n;
      sn->done = 1;
    }
    for (int i = 0; i < 16; i++) { trace_buf[trace_n].hash[i] = 0; trace_buf[trace_n].bytes[i] = 0; }
    if (hash_all) for (int i = 0; i < k->nargs && i < 16; i++) if (argptr[i]) { trace_buf[trace_n].hash[i] = hash_bytes(argptr[i], argbytes[i]); trace_buf[trace_n].bytes[i] = argbytes[i]; }
    trace_n++;
    /* the same launch again, up to repeat_extra times (rdcl_set_repeat): words of its buffer arguments each extra launch changes */
    if (repeat_extra > 0 && occ == repeat_occ && !strcmp(full, repeat_name) && !skip_launches) {
      void *before[MAXARGS];
      for (int i = 0; i < k->nargs; i++) before[i] = argptr[i] ? malloc(argbytes[i]) : NULL;
      for (int e = 0; e < repeat_extra; e++) {
        for (int i = 0; i < k->nargs; i++) if (argptr[i]) memcpy(before[i], argptr[i], argbytes[i]);
        run_items(k, ia, fa, g0, g1);
        int changed = 0;
        for (int i = 0; i < k->nargs; i++) if (argptr[i]) {
          const uint32_t *x = (const uint32_t *)before[i], *y = (const uint32_t *)argptr[i];
          for (size_t w = 0; w < argbytes[i] / 4; w++) changed += x[w] != y[w];
        }
        repeat_changed[e] = changed;
        if (changed == 0) break;
      }
      for (int i = 0; i < k->nargs; i++) free(before[i]);
    }
  }

  if (ev) { *ev = (cl_event)calloc(1, sizeof(struct _cl_event)); (*ev)->refs = 1; }
  return CL_SUCCESS;
}

/* ---------------------------------------------------------------- buffers */

cl_mem clCreateBuffer(cl_context c, cl_mem_flags flags, size_t size, void *host, cl_int *err) {
  (void)c;
  cl_mem m = (cl_mem)calloc(1, sizeof(*m));
  m->magic = MEM_MAGIC;
  m->size = size;
  m->data = calloc(1, size ? size : 1);
  if ((flags & (CL_MEM_COPY_HOST_PTR | CL_MEM_USE_HOST_PTR)) && host) memcpy(m->data, host, size);
  m->next = live_mems;
  if (live_mems) live_mems->prev = m;
  live_mems = m;
  if (err) *err = CL_SUCCESS;
  return m;
}

cl_int clReleaseMemObject(cl_mem m) {
  if (!m || m->magic != MEM_MAGIC) return CL_INVALID_MEM_OBJECT;
  if (m->prev) m->prev->next = m->next; else live_mems = m->next;
  if (m->next) m->next->prev = m->prev;
  m->magic = 0;
  free(m->data);
  free(m);
  return CL_SUCCESS;
}

static cl_event new_event(void) { cl_event e = (cl_event)calloc(1, sizeof(*e)); e->refs = 1; return e; }

cl_int clEnqueueReadBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, size_t off, size_t size, void *dst,
                           cl_uint nev, const cl_event *evl, cl_event *ev) {
  (void)q; (void)blocking; (void)nev; (void)evl;
  if (!m || m->magic != MEM_MAGIC || off + size > m->size) return CL_INVALID_VALUE;
  if (subst_at < subst_n) {      /* rdcl_substitute_reads: the caller's bytes in place of the buffer's (what lies beyond them: the buffer's) */
    const size_t have = subst_size[subst_at] < size ? subst_size[subst_at] : size;
    memcpy(dst, subst_data[subst_at], have);
    if (have < size) memcpy((char *)dst + have, (char *)m->data + off + have, size - have);
    subst_at++;
  } else
  memcpy(dst, (char *)m->data + off, size);
  if (ev) *ev = new_event();
  return CL_SUCCESS;
}

cl_int clEnqueueWriteBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, size_t off, size_t size, const void *src,
                            cl_uint nev, const cl_event *evl, cl_event *ev) {
  (void)q; (void)blocking; (void)nev; (void)evl;
  if (!m || m->magic != MEM_MAGIC || off + size > m->size) return CL_INVALID_VALUE;
  memcpy((char *)m->data + off, src, size);
  if (ev) *ev = new_event();
  return CL_SUCCESS;
}

void *clEnqueueMapBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, cl_map_flags flags, size_t off, size_t size,
                         cl_uint nev, const cl_event *evl, cl_event *ev, cl_int *err) {
  (void)q; (void)blocking; (void)flags; (void)size; (void)nev; (void)evl;
  if (ev) *ev = new_event();
  if (err) *err = CL_SUCCESS;
  return (char *)m->data + off;
}

cl_int clEnqueueUnmapMemObject(cl_command_queue q, cl_mem m, void *ptr, cl_uint nev, const cl_event *evl, cl_event *ev) {
  (void)q; (void)m; (void)ptr; (void)nev; (void)evl;
  if (ev) *ev = new_event();
  return CL_SUCCESS;
}

/* ---------------------------------------------------------------- events */

cl_int clGetEventInfo(cl_event e, cl_event_info what, size_t sz, void *val, size_t *ret) {
  (void)e;
  if (what != CL_EVENT_COMMAND_EXECUTION_STATUS || sz < sizeof(cl_int)) return CL_INVALID_VALUE;
  *(cl_int *)val = CL_COMPLETE;
  if (ret) *ret = sizeof(cl_int);
  return CL_SUCCESS;
}

cl_int clGetEventProfilingInfo(cl_event e, cl_profiling_info what, size_t sz, void *val, size_t *ret) {
  (void)e; (void)what;
  if (sz < sizeof(cl_ulong)) return CL_INVALID_VALUE;
  *(cl_ulong *)val = 0;
  if (ret) *ret = sizeof(cl_ulong);
  return CL_SUCCESS;
}

cl_int clRetainEvent(cl_event e) { if (e) e->refs++; return CL_SUCCESS; }
cl_int clReleaseEvent(cl_event e) { if (e && --e->refs == 0) free(e); return CL_SUCCESS; }
