/* libm_sweep.c -- exhaustive comparison of oracle/glibc_flt32.h with the libm of the machine it runs on.
 *
 *   libm_sweep <fn> <first_bits_hex> <last_bits_hex> [y] [threads]
 *
 * fn: cosf | sinf | log10f | logf | expf | powf (powf needs y).  Every binary32 whose bit pattern lies in
 * [first, last] is evaluated by both; NaN results compare equal to each other.  Prints one line
 *   "<fn> n=<count> mismatches=<m>" followed by up to 16 "x=%08x mine=%08x libm=%08x" lines; exit code 0 iff m == 0.
 * Built by tests/test_libm_pin.py: gcc -O2 -ffp-contract=off -mfma -pthread libm_sweep.c -lm */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../oracle/glibc_flt32.h"

typedef struct { int fn; uint64_t first, last; float y; uint64_t bad; uint32_t ex[16][3]; int nex; } job_t;

static float mine(int fn, float x, float y) {
    switch (fn) {
    case 0: return gl_cosf(x);
    case 1: return gl_sinf(x);
    case 2: return gl_log10f(x);
    case 3: return gl_logf(x);
    case 4: return gl_expf(x);
    default: return gl_powf(x, y);
    }
}
static float theirs(int fn, float x, float y) {
    switch (fn) {
    case 0: return cosf(x);
    case 1: return sinf(x);
    case 2: return log10f(x);
    case 3: return logf(x);
    case 4: return expf(x);
    default: return powf(x, y);
    }
}
static void *run(void *arg) {
    job_t *j = (job_t *)arg;
    for (uint64_t b = j->first; b <= j->last; b++) {
        const float x = gl_asfloat((uint32_t)b);
        volatile float xv = x;                       /* keep gcc from folding the libm call */
        const float a = mine(j->fn, x, j->y), c = theirs(j->fn, xv, j->y);
        const uint32_t ua = gl_asuint(a), uc = gl_asuint(c);
        if (ua != uc && !(a != a && c != c)) {
            if (j->nex < 16) { j->ex[j->nex][0] = (uint32_t)b; j->ex[j->nex][1] = ua; j->ex[j->nex][2] = uc; j->nex++; }
            j->bad++;
        }
    }
    return 0;
}
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s fn first last [y] [threads]\n", argv[0]); return 2; }
    static const char *names[] = {"cosf", "sinf", "log10f", "logf", "expf", "powf"};
    int fn = -1;
    for (int i = 0; i < 6; i++) if (!strcmp(argv[1], names[i])) fn = i;
    if (fn < 0) return 2;
    const uint64_t first = strtoull(argv[2], 0, 16), last = strtoull(argv[3], 0, 16);
    const float y = argc > 4 ? strtof(argv[4], 0) : 0.0f;
    int nt = argc > 5 ? atoi(argv[5]) : 8;
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    if (last < first || last > 0xffffffffull) return 2;
    const uint64_t n = last - first + 1;
    job_t *jobs = calloc((size_t)nt, sizeof(job_t));
    pthread_t *th = calloc((size_t)nt, sizeof(pthread_t));
    for (int t = 0; t < nt; t++) {
        jobs[t].fn = fn; jobs[t].y = y;
        jobs[t].first = first + n * (uint64_t)t / (uint64_t)nt;
        jobs[t].last = first + n * (uint64_t)(t + 1) / (uint64_t)nt - 1;
        if (jobs[t].last + 1 == jobs[t].first) { jobs[t].first = 1; jobs[t].last = 0; }   /* empty slice */
        pthread_create(&th[t], 0, run, &jobs[t]);
    }
    uint64_t bad = 0;
    for (int t = 0; t < nt; t++) { pthread_join(th[t], 0); bad += jobs[t].bad; }
    printf("%s n=%llu mismatches=%llu\n", names[fn], (unsigned long long)n, (unsigned long long)bad);
    int shown = 0;
    for (int t = 0; t < nt && shown < 16; t++)
        for (int e = 0; e < jobs[t].nex && shown < 16; e++, shown++)
            printf("x=%08x mine=%08x libm=%08x\n", jobs[t].ex[e][0], jobs[t].ex[e][1], jobs[t].ex[e][2]);
    return bad ? 1 : 0;
}
