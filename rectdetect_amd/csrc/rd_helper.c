/*
 * rectdetect-mi355x: helper.h utilities (reference helper.c), written from the behaviour described there.
 * The ArrayMap keeps the reference's bucket function and insertion order inside a bucket because the order of
 * ArrayMap_keyArray() decides the order in which the detector reports rectangles (reference oclrect.c:1100-1103).
 */
#define _GNU_SOURCE
#include <assert.h>
#include <ctype.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#include "helper.h"

void exitf(int code, const char *mes, ...) {
  va_list ap;
  va_start(ap, mes);
  vfprintf(stderr, mes, ap);
  va_end(ap);
  fflush(stderr);
  exit(code);
}

char *readFileAsStr(const char *fn, int maxSize) {
  FILE *fp = fopen(fn, "r");
  if (!fp) exitf(-1, "Couldn't open file %s\n", fn);
  fseek(fp, 0, SEEK_END);
  long size = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  if (size > maxSize) exitf(-1, "readFileAsStr : file too large (%d bytes)\n", (int)size);
  char *buf = (char *)malloc((size_t)size + 10);
  if (!buf) exitf(-1, "readFileAsStr : malloc failed\n");
  size = (long)fread(buf, 1, (size_t)size, fp);
  buf[size] = '\0';
  fclose(fp);
  return buf;
}

char *readFileAsStrN(const char **fn) {
  size_t total = 0;
  char *buf = (char *)malloc(10);
  buf[0] = '\0';
  for (int i = 0; fn[i] != NULL; i++) {
    char *one = readFileAsStr(fn[i], 1000000);
    size_t l = strlen(one);
    if (total + l > 1000000) exitf(-1, "readFileAsStrN : total file size %d bytes is too large\n", (int)(total + l));
    buf = (char *)realloc(buf, total + l + 10);
    if (!buf) exitf(-1, "readFileAsStrN : realloc failed\n");
    memcpy(buf + total, one, l + 1);
    total += l;
    free(one);
  }
  return buf;
}

void String_trim(char *str) {
  char *src = str, *dst = str, *end = str;
  while (*src && isspace((unsigned char)*src)) src++;
  for (; *src; src++) {
    *dst++ = *src;
    if (!isspace((unsigned char)*src)) end = dst;
  }
  *end = '\0';
}

int64_t currentTimeMillis() {
  struct timeval tp;
  gettimeofday(&tp, NULL);
  return tp.tv_sec * (int64_t)1000 + tp.tv_usec / 1000;
}

void sleepMillis(int ms) { usleep((useconds_t)ms * 1000); }

/* ---- ArrayMap: 1024 buckets, bucket = xor of the four 10-bit groups of the key (reference helper.c:127-134),
 * entries appended per bucket, removal moves the bucket's last entry into the hole (helper.c:192-212). */
#define AM_BITS 10
#define AM_NB (1 << AM_BITS)
#define AM_MAGIC 0x8693bd21u

typedef struct { uint64_t key; void *value; } am_node;
struct ArrayMap { uint32_t magic; am_node *b[AM_NB]; int n[AM_NB], cap[AM_NB], total; };

static int am_bucket(uint64_t k) { return (int)((k ^ (k >> AM_BITS) ^ (k >> (AM_BITS * 2)) ^ (k >> (AM_BITS * 3))) & (AM_NB - 1)); }

ArrayMap *initArrayMap() {
  ArrayMap *m = (ArrayMap *)calloc(1, sizeof(ArrayMap));
  m->magic = AM_MAGIC;
  return m;
}

void ArrayMap_dispose(ArrayMap *m) {
  assert(m && m->magic == AM_MAGIC);
  for (int i = 0; i < AM_NB; i++) free(m->b[i]);
  m->magic = 0;
  free(m);
}

int ArrayMap_size(ArrayMap *m) { assert(m && m->magic == AM_MAGIC); return m->total; }

void *ArrayMap_get(ArrayMap *m, uint64_t key) {
  assert(m && m->magic == AM_MAGIC);
  const int h = am_bucket(key);
  for (int i = 0; i < m->n[h]; i++) if (m->b[h][i].key == key) return m->b[h][i].value;
  return NULL;
}

void *ArrayMap_remove(ArrayMap *m, uint64_t key) {
  assert(m && m->magic == AM_MAGIC);
  const int h = am_bucket(key);
  for (int i = 0; i < m->n[h]; i++)
    if (m->b[h][i].key == key) {
      void *old = m->b[h][i].value;
      m->b[h][i] = m->b[h][m->n[h] - 1];
      m->n[h]--; m->total--;
      return old;
    }
  return NULL;
}

void *ArrayMap_put(ArrayMap *m, uint64_t key, void *value) {
  if (value == NULL) return ArrayMap_remove(m, key);
  assert(m && m->magic == AM_MAGIC);
  const int h = am_bucket(key);
  for (int i = 0; i < m->n[h]; i++)
    if (m->b[h][i].key == key) { void *old = m->b[h][i].value; m->b[h][i].value = value; return old; }
  if (m->n[h] >= m->cap[h]) {
    m->cap[h] = m->cap[h] ? m->cap[h] * 2 : 8;
    m->b[h] = (am_node *)realloc(m->b[h], (size_t)m->cap[h] * sizeof(am_node));
  }
  m->b[h][m->n[h]].key = key;
  m->b[h][m->n[h]].value = value;
  m->n[h]++; m->total++;
  return NULL;
}

uint64_t *ArrayMap_keyArray(ArrayMap *m) {
  assert(m && m->magic == AM_MAGIC);
  uint64_t *a = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m->total ? m->total : 1));
  int p = 0;
  for (int h = 0; h < AM_NB; h++) for (int i = 0; i < m->n[h]; i++) a[p++] = m->b[h][i].key;
  return a;
}

void **ArrayMap_valueArray(ArrayMap *m) {
  assert(m && m->magic == AM_MAGIC);
  void **a = (void **)malloc(sizeof(void *) * (size_t)(m->total ? m->total : 1));
  int p = 0;
  for (int h = 0; h < AM_NB; h++) for (int i = 0; i < m->n[h]; i++) a[p++] = m->b[h][i].value;
  return a;
}

uint64_t ArrayMap_getKey(ArrayMap *m, int idx) {
  assert(m && m->magic == AM_MAGIC);
  for (int h = 0; h < AM_NB; h++) { if (idx < m->n[h]) return m->b[h][idx].key; idx -= m->n[h]; }
  return 0;
}

void *ArrayMap_getValue(ArrayMap *m, int idx) {
  assert(m && m->magic == AM_MAGIC);
  for (int h = 0; h < AM_NB; h++) { if (idx < m->n[h]) return m->b[h][idx].value; idx -= m->n[h]; }
  return NULL;
}
