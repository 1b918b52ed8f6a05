/*
 * TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product library.
 *
 * An OBSERVER on top of a REAL OpenCL implementation: linked into oracle/_ref/librdref_ocl.so in front of the system's OpenCL loader, it
 * forwards the five entry points it wraps to the real library untouched and only watches: which kernel of which of the reference's three
 * programs is launched, with which buffer arguments, and - on request - reads a buffer back right after a chosen launch.  It offers the
 * snapshot interface of the serial stand-in (rdcl_device.c: rdcl_trace_*, rdcl_snapshot_*), so that the planes the reference produces stage by
 * stage ON THE DEVICE (the GPU box: the MI355X through ROCm's OpenCL, the reference's .cl sources built by the vendor's compiler) can be
 * compared with the oracle's, launch by launch (tools/ref_stages_on_opencl.py).  Nothing here computes anything.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CL_TARGET_OPENCL_VERSION 120
#define CL_USE_DEPRECATED_OPENCL_1_2_APIS
#include <CL/cl.h>

#define MAXK 512
#define MAXARGS 24
#define MAXP 16

static void *real(const char *name) {
  static void *lib = NULL;
  if (!lib) lib = dlopen("libOpenCL.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("libOpenCL.so", RTLD_NOW | RTLD_GLOBAL);
  void *f = lib ? dlsym(lib, name) : NULL;
  if (!f) { fprintf(stderr, "rdcl_observe: %s not found in the system's OpenCL loader\n", name); abort(); }
  return f;
}

static struct { cl_program p; int tag; } progs[MAXP];
static int nprogs = 0;
static const char *tag_name[3] = { "imgutil", "polyline", "rect" };
static struct { cl_kernel k; int tag; char name[64]; cl_mem arg[MAXARGS]; } kern[MAXK];
static int nkern = 0;

typedef struct { char name[96]; uint64_t hash[MAXARGS]; size_t bytes[MAXARGS]; } launch_t;
typedef struct { char name[96]; int occurrence, argidx, ordinal; void *data; size_t size; int done; } snap_t;
static launch_t *trace_buf = NULL;
static int trace_n = 0, trace_cap = 0;
static snap_t snaps[256];
static int nsnaps = 0;

void rdcl_trace_reset(void) { trace_n = 0; }
int rdcl_trace_count(void) { return trace_n; }
const char *rdcl_trace_name(int i) { return (i >= 0 && i < trace_n) ? trace_buf[i].name : ""; }
/* rdcl_hash_all(1): every launch also records a fingerprint (the stand-in's function, rdcl_device.c) and the size of each of its buffer arguments, read back right after it.
 * rdcl_zero_fill(1): buffers the reference creates without contents are filled with zeros before the first launch after their creation - what the serial stand-in's
 *   buffers start as; the reference reads planes it never wrote, so without it the device's stale memory takes part.  The only thing this observer ever
 *   writes, and only on request.
 * rdcl_snapshot_limit(bytes): snapshots keep at most so many bytes of a buffer (0: all of it). */
static int hash_all = 0, zero_fill = 0;
static size_t snap_limit = 0;
void rdcl_hash_all(int on) { hash_all = on; }
void rdcl_zero_fill(int on) { zero_fill = on; }
void rdcl_snapshot_limit(size_t bytes) { snap_limit = bytes; }
uint64_t rdcl_trace_hash(int i, int arg) { return (i >= 0 && i < trace_n && arg >= 0 && arg < MAXARGS) ? trace_buf[i].hash[arg] : 0; }
size_t rdcl_trace_bytes(int i, int arg) { return (i >= 0 && i < trace_n && arg >= 0 && arg < MAXARGS) ? trace_buf[i].bytes[arg] : 0; }
static uint64_t hash_bytes(const void *p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  const uint64_t *w = (const uint64_t *)p;
  for (size_t i = 0; i < n / 8; i++) { h ^= w[i]; h *= 1099511628211ull; }
  const unsigned char *b = (const unsigned char *)p + (n & ~(size_t)7);
  for (size_t i = 0; i < (n & 7); i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
/* the buffers the reference created (so that an 8-byte scalar argument is never mistaken for one), and which of them still wait for their zeros */
#define MAXMEM 4096
static struct { cl_mem m; size_t bytes; int unfilled; } mems[MAXMEM];
static int nmems = 0;
static int mem_index(cl_mem m) { for (int i = 0; i < nmems; i++) if (mems[i].m == m) return i; return -1; }

cl_mem clCreateBuffer(cl_context ctx, cl_mem_flags flags, size_t size, void *host, cl_int *err) {
  cl_mem (*f)(cl_context, cl_mem_flags, size_t, void *, cl_int *) = real("clCreateBuffer");
  cl_mem m = f(ctx, flags, size, host, err);
  if (m) {
    int i = mem_index(NULL);
    if (i < 0 && nmems < MAXMEM) i = nmems++;
    if (i >= 0) { mems[i].m = m; mems[i].bytes = size; mems[i].unfilled = !(flags & (CL_MEM_COPY_HOST_PTR | CL_MEM_USE_HOST_PTR)); }
  }
  return m;
}
cl_int clReleaseMemObject(cl_mem m) {
  cl_int (*f)(cl_mem) = real("clReleaseMemObject");
  const int i = mem_index(m);
  if (i >= 0) mems[i].m = NULL;
  return f(m);
}
/* (the reference's "pinned" host memory is a CL_MEM_ALLOC_HOST_PTR buffer mapped for good, oclhelper.c:837-851, and its contents become those of the io buffers: it gets its
 *  zeros through the mapping - a mapped buffer must not be filled by the device) */
void *clEnqueueMapBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, cl_map_flags flags, size_t off, size_t size, cl_uint nev, const cl_event *evl, cl_event *ev, cl_int *err) {
  void *(*f)(cl_command_queue, cl_mem, cl_bool, cl_map_flags, size_t, size_t, cl_uint, const cl_event *, cl_event *, cl_int *) = real("clEnqueueMapBuffer");
  void *p = f(q, m, blocking, flags, off, size, nev, evl, ev, err);
  const int i = mem_index(m);
  if (p && i >= 0 && mems[i].unfilled) {
    if (zero_fill && blocking && (flags & CL_MAP_WRITE)) memset(p, 0, size);
    mems[i].unfilled = 0;
  }
  return p;
}
/* a buffer's bytes on the host: directly, or - the reference makes its planes CL_MEM_HOST_NO_ACCESS - through a device-side copy into a buffer the host may read */
static cl_int read_back(cl_command_queue q, cl_mem m, size_t bytes, void *dst) {
  cl_int (*rd)(cl_command_queue, cl_mem, cl_bool, size_t, size_t, void *, cl_uint, const cl_event *, cl_event *) = real("clEnqueueReadBuffer");
  cl_int re = rd(q, m, CL_TRUE, 0, bytes, dst, 0, NULL, NULL);
  if (re != CL_INVALID_OPERATION) return re;
  static cl_mem stage = NULL; static size_t stage_bytes = 0; static cl_context stage_ctx = NULL;
  cl_int (*info)(cl_mem, cl_mem_info, size_t, void *, size_t *) = real("clGetMemObjectInfo");
  cl_mem (*mk)(cl_context, cl_mem_flags, size_t, void *, cl_int *) = real("clCreateBuffer");
  cl_int (*cp)(cl_command_queue, cl_mem, cl_mem, size_t, size_t, size_t, cl_uint, const cl_event *, cl_event *) = real("clEnqueueCopyBuffer");
  cl_int (*rel)(cl_mem) = real("clReleaseMemObject");
  cl_context ctx = NULL;
  cl_int ce = info(m, CL_MEM_CONTEXT, sizeof(ctx), &ctx, NULL);
  if (ce != CL_SUCCESS) return ce;
  if (!stage || stage_bytes < bytes || stage_ctx != ctx) {
    if (stage) rel(stage);
    stage = mk(ctx, CL_MEM_READ_WRITE, bytes, NULL, &ce);
    if (!stage || ce != CL_SUCCESS) { stage = NULL; return ce; }
    stage_bytes = bytes; stage_ctx = ctx;
  }
  re = cp(q, m, stage, 0, 0, bytes, 0, NULL, NULL);
  if (re == CL_SUCCESS) re = rd(q, stage, CL_TRUE, 0, bytes, dst, 0, NULL, NULL);
  return re;
}
void rdcl_snapshot_clear(void) { for (int i = 0; i < nsnaps; i++) free(snaps[i].data); nsnaps = 0; }
/* a copy of buffer argument `argidx` of the `occurrence`-th (0-based, since the last rdcl_trace_reset) launch of "<program>:<kernel>", taken right after it */
int rdcl_snapshot_request(const char *name, int occurrence, int argidx) {
  if (nsnaps >= 256) return -1;
  snap_t *s = &snaps[nsnaps];
  memset(s, 0, sizeof(*s));
  snprintf(s->name, sizeof(s->name), "%s", name);
  s->occurrence = occurrence; s->argidx = argidx; s->ordinal = -1;
  return nsnaps++;
}
int rdcl_snapshot_fetch(int h, void **data, size_t *size, int *ordinal) {
  if (h < 0 || h >= nsnaps || !snaps[h].done) return -1;
  *data = snaps[h].data; *size = snaps[h].size;
  if (ordinal) *ordinal = snaps[h].ordinal;
  return 0;
}

cl_program clCreateProgramWithSource(cl_context ctx, cl_uint count, const char **strings, const size_t *lengths, cl_int *err) {
  cl_program (*f)(cl_context, cl_uint, const char **, const size_t *, cl_int *) = real("clCreateProgramWithSource");
  cl_program p = f(ctx, count, strings, lengths, err);
  int tag = 0;      /* which of the reference's three sources this is: by a kernel only that file defines */
  for (cl_uint i = 0; i < count; i++) {
    if (!strings[i]) continue;
    if (strstr(strings[i], "labelMergeMain")) tag = 2;
    else if (strstr(strings[i], "refine_pass3")) tag = 1;
  }
  if (nprogs < MAXP) { progs[nprogs].p = p; progs[nprogs].tag = tag; nprogs++; }
  return p;
}

cl_kernel clCreateKernel(cl_program program, const char *name, cl_int *err) {
  cl_kernel (*f)(cl_program, const char *, cl_int *) = real("clCreateKernel");
  cl_kernel k = f(program, name, err);
  if (k && nkern < MAXK) {
    int tag = 0;
    for (int i = 0; i < nprogs; i++) if (progs[i].p == program) tag = progs[i].tag;
    memset(&kern[nkern], 0, sizeof(kern[nkern]));
    kern[nkern].k = k; kern[nkern].tag = tag;
    snprintf(kern[nkern].name, sizeof(kern[nkern].name), "%s", name);
    nkern++;
  }
  return k;
}

cl_int clSetKernelArg(cl_kernel kernel, cl_uint idx, size_t size, const void *value) {
  cl_int (*f)(cl_kernel, cl_uint, size_t, const void *) = real("clSetKernelArg");
  if (size == sizeof(cl_mem) && value && idx < MAXARGS)
    for (int i = 0; i < nkern; i++) if (kern[i].k == kernel) kern[i].arg[idx] = *(const cl_mem *)value;      /* (8-byte scalars land here too: only ever read back on request, checked then) */
  return f(kernel, idx, size, value);
}

cl_int clEnqueueNDRangeKernel(cl_command_queue q, cl_kernel kernel, cl_uint dim, const size_t *off, const size_t *gws, const size_t *lws, cl_uint nev, const cl_event *evs, cl_event *ev) {
  cl_int (*f)(cl_command_queue, cl_kernel, cl_uint, const size_t *, const size_t *, const size_t *, cl_uint, const cl_event *, cl_event *) = real("clEnqueueNDRangeKernel");
  if (zero_fill) {
    cl_int (*fill)(cl_command_queue, cl_mem, const void *, size_t, size_t, size_t, cl_uint, const cl_event *, cl_event *) = real("clEnqueueFillBuffer");
    static const cl_int zero = 0;
    for (int i = 0; i < nmems; i++) if (mems[i].m && mems[i].unfilled) {
      const size_t b = mems[i].bytes & ~(size_t)3;
      cl_int fe = b ? fill(q, mems[i].m, &zero, sizeof(zero), 0, b, 0, NULL, NULL) : CL_SUCCESS;
      if (fe != CL_SUCCESS) fprintf(stderr, "rdcl_observe: zero fill of a %zu-byte buffer failed (%d)\n", mems[i].bytes, (int)fe);
      mems[i].unfilled = 0;
    }
  }
  const cl_int ret = f(q, kernel, dim, off, gws, lws, nev, evs, ev);
  int ki = -1;
  for (int i = 0; i < nkern; i++) if (kern[i].k == kernel) ki = i;
  if (ki < 0) return ret;
  char full[96];
  snprintf(full, sizeof(full), "%s:%s", tag_name[kern[ki].tag], kern[ki].name);
  int occ = 0;
  for (int i = 0; i < trace_n; i++) if (!strcmp(trace_buf[i].name, full)) occ++;
  if (trace_n == trace_cap) { trace_cap = trace_cap ? trace_cap * 2 : 1024; trace_buf = realloc(trace_buf, (size_t)trace_cap * sizeof(*trace_buf)); }
  snprintf(trace_buf[trace_n].name, sizeof(trace_buf[trace_n].name), "%s", full);
  memset(trace_buf[trace_n].hash, 0, sizeof(trace_buf[trace_n].hash)); memset(trace_buf[trace_n].bytes, 0, sizeof(trace_buf[trace_n].bytes));
  if (hash_all) {
    cl_int (*fin)(cl_command_queue) = real("clFinish");
    static void *host = NULL; static size_t host_bytes = 0;
    fin(q);
    for (int a = 0; a < MAXARGS; a++) {
      const int mi = kern[ki].arg[a] ? mem_index(kern[ki].arg[a]) : -1;
      if (mi < 0) continue;
      const size_t b = mems[mi].bytes;
      if (host_bytes < b) { free(host); host = malloc(b); host_bytes = host ? b : 0; }
      if (!host) continue;
      if (read_back(q, mems[mi].m, b, host) == CL_SUCCESS) { trace_buf[trace_n].hash[a] = hash_bytes(host, b); trace_buf[trace_n].bytes[a] = b; }
    }
  }
  const int ordinal = trace_n++;
  for (int s = 0; s < nsnaps; s++) {
    snap_t *sn = &snaps[s];
    if (sn->done || strcmp(sn->name, full)) continue;
    if (!((sn->occurrence >= 0 && sn->occurrence == occ) || (sn->occurrence < 0 && -sn->occurrence - 1 == ordinal))) continue;
    cl_mem m = (sn->argidx >= 0 && sn->argidx < MAXARGS) ? kern[ki].arg[sn->argidx] : NULL;
    size_t bytes = 0;
    cl_int (*info)(cl_mem, cl_mem_info, size_t, void *, size_t *) = real("clGetMemObjectInfo");
    cl_int ie = m ? info(m, CL_MEM_SIZE, sizeof(bytes), &bytes, NULL) : -1;
    if (getenv("RDCL_OBSERVE_DEBUG")) fprintf(stderr, "rdcl_observe: %s occurrence %d arg %d: mem %p info %d bytes %zu\n", full, occ, sn->argidx, (void *)m, (int)ie, bytes);
    if (!m || ie != CL_SUCCESS || bytes == 0) continue;
    cl_int (*fin)(cl_command_queue) = real("clFinish");
    fin(q);
    if (snap_limit && bytes > snap_limit) bytes = snap_limit;
    sn->data = malloc(bytes);
    cl_int re = sn->data ? read_back(q, m, bytes, sn->data) : -1;
    if (getenv("RDCL_OBSERVE_DEBUG")) fprintf(stderr, "rdcl_observe:   read %d\n", (int)re);
    if (re == CL_SUCCESS) { sn->size = bytes; sn->ordinal = ordinal; sn->done = 1; }
  }
  return ret;
}
