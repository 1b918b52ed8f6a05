/*
 * rectdetect-mi355x: utility surface of the reference's helper.h (reference helper.h:12-31),
 * re-implemented in rectdetect_amd/csrc/rd_helper.c.  Same names, argument meaning and error
 * behaviour (exitf prints to stderr and exits; nothing returns an error code).
 */
#ifndef RD_COMPAT_HELPER_H
#define RD_COMPAT_HELPER_H
#include <stdint.h>
#if defined(__cplusplus)
extern "C" {
#endif

/* print a printf-style message to stderr and exit(code)          (reference helper.c:31-38) */
void exitf(int code, const char *mes, ...);
/* whole file as a NUL-terminated malloc'd string, fatal on error  (reference helper.c:40-62) */
char *readFileAsStr(const char *fn, int maxSize);
/* concatenation of a NULL-terminated list of files               (reference helper.c:64-90) */
char *readFileAsStrN(const char **fn);
/* wall clock in milliseconds / sleep                              (reference helper.c:105-125) */
int64_t currentTimeMillis();
void sleepMillis(int ms);
/* strip leading and trailing white space in place                 (reference helper.c:92-103) */
void String_trim(char *str);

/* uint64 -> pointer map; key order of ArrayMap_keyArray is part of the observable behaviour of the
 * detector (it fixes the order of the returned rectangles)        (reference helper.c:127-267) */
typedef struct ArrayMap ArrayMap;
ArrayMap *initArrayMap();
void ArrayMap_dispose(ArrayMap *thiz);
int ArrayMap_size(ArrayMap *thiz);
void *ArrayMap_remove(ArrayMap *thiz, uint64_t key);
void *ArrayMap_put(ArrayMap *thiz, uint64_t key, void *value);
void *ArrayMap_get(ArrayMap *thiz, uint64_t key);
uint64_t *ArrayMap_keyArray(ArrayMap *thiz);
void **ArrayMap_valueArray(ArrayMap *thiz);
uint64_t ArrayMap_getKey(ArrayMap *thiz, int idx);
void *ArrayMap_getValue(ArrayMap *thiz, int idx);

#if defined(__cplusplus)
}
#endif
#endif
