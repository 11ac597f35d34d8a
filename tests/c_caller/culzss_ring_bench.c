/* Throughput of the reference's host-pointer CULZSS ABI (include/culzss.h) from a plain-C caller, PCIe included.
 *
 * The reference's callers keep a ring of four slots in flight (cuda-lzss-cluster/culzss.c:85-176: a producer fills a
 * slot, the gpu_consumer thread calls compression_kernel_wrapper on it -- asynchronous by contract,
 * gpu_compress.cu:352-460 -- the cpu_consumer waits with onestream_finish_GPU and runs aftercompression_wrapper, a
 * writer drains it).  Two passes over the same `nbuf` buffers:
 *   seq   one buffer at a time: wrapper, finish, aftercompression (nothing overlaps: rounds 1-4's figure);
 *   ring  four slots in flight, ONE caller thread: slot s is retired (finish + aftercompression) right before it is
 *         refilled, so the H2D of buffer k + 3, the kernels of k + 1 .. k + 2 and the D2H of k overlap;
 *   threads  the reference's own shape: a producer thread fills slots, a GPU thread launches them, a CPU thread
 *         retires them, a ledger of slot states between them (culzss.c's queue).
 * All passes must produce the same bytes (sizes and an FNV-1a hash of every packed buffer, in order).
 * Built with gcc, no oracle: bench.py compiles and runs it for `encode_with_pcie_staging_GBps`.
 *   usage: culzss_ring_bench [nbuf = 256] [distinct buffers = 16]   -> one line of key=value pairs */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include "culzss.h"

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void log_text(unsigned char *p, int n, uint32_t seed)
{   /* log-style lines: repetitive structure, varying fields (the generator of culzss_rig.c) */
    static const char *lvl[] = {"INFO", "WARN", "DEBUG", "ERROR"};
    int o = 0;
    while (o < n) {
        char line[160];
        const int k = snprintf(line, sizeof line, "2026-09-28T%02u:%02u:%02u.%03uZ host-%02u svc-%c[%u]: %s request id=%06x latency=%ums\n",
                               lcg(&seed) % 24, lcg(&seed) % 60, lcg(&seed) % 60, lcg(&seed) % 1000, lcg(&seed) % 16,
                               'a' + (char)(lcg(&seed) % 5), 1000 + lcg(&seed) % 9000, lvl[lcg(&seed) % 4], lcg(&seed) & 0xFFFFFF,
                               lcg(&seed) % 500);
        const int c = k < n - o ? k : n - o;
        memcpy(p + o, line, (size_t)c);
        o += c;
    }
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint64_t fnv(uint64_t h, const unsigned char *p, int n)
{
    for (int i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

#define BUF (1 << 20)
#define NSLOT 4

static unsigned char *buf[NSLOT], *bufout[NSLOT], *in_d, *out_d, **pool;
static int npool;

/* retire slot s: what culzss.c's cpu_consumer does; folds the packed bytes into the pass's hash */
static int retire(int s, uint64_t *h, long long *bytes, int hash)
{
    int comp = 0;
    if (onestream_finish_GPU(s) != 1) return 0;
    const int ok = aftercompression_wrapper(buf[s], BUF, bufout[s], &comp);
    if (!ok) comp = BUF;                                       /* stored raw: the slot still holds the input */
    *bytes += comp;
    if (hash) *h = fnv(*h ^ (uint64_t)comp, buf[s], comp);
    return 1;
}

static double pass(int nbuf, int inflight, uint64_t *h, long long *bytes, int hash)
{
    *h = 1469598103934665603ull; *bytes = 0;
    const double t0 = now();
    for (int i = 0; i < nbuf + inflight; i++) {
        const int s = i % inflight;
        if (i >= inflight && !retire(s, h, bytes, hash)) return -1.0;
        if (i < nbuf) {
            memcpy(buf[s], pool[i % npool], (size_t)BUF);      /* the producer: the next 1 MiB of the file */
            if (compression_kernel_wrapper(buf[s], BUF, bufout[s], 0, 0, 128, 0, s, in_d, out_d) != 1) return -1.0;
        }
    }
    return now() - t0;
}

/* ---- the three-thread pipeline: slot states FREE -> FILLED -> LAUNCHED -> FREE, buffers in order ---- */
enum { FREE = 0, FILLED = 1, LAUNCHED = 2 };
static pthread_mutex_t q_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t q_cv = PTHREAD_COND_INITIALIZER;
static int q_state[NSLOT], q_nbuf, q_hash, q_fail;
static uint64_t q_h;
static long long q_bytes;

static void q_wait(int s, int want)
{
    pthread_mutex_lock(&q_mu);
    while (q_state[s] != want && !q_fail) pthread_cond_wait(&q_cv, &q_mu);
    pthread_mutex_unlock(&q_mu);
}
static void q_set(int s, int v)
{
    pthread_mutex_lock(&q_mu);
    q_state[s] = v;
    pthread_cond_broadcast(&q_cv);
    pthread_mutex_unlock(&q_mu);
}
static void q_abort(void) { pthread_mutex_lock(&q_mu); q_fail = 1; pthread_cond_broadcast(&q_cv); pthread_mutex_unlock(&q_mu); }

static void *t_producer(void *a)
{
    (void)a;
    for (int i = 0; i < q_nbuf && !q_fail; i++) {
        const int s = i % NSLOT;
        q_wait(s, FREE);
        memcpy(buf[s], pool[i % npool], (size_t)BUF);
        q_set(s, FILLED);
    }
    return NULL;
}
static void *t_gpu(void *a)
{
    (void)a;
    for (int i = 0; i < q_nbuf && !q_fail; i++) {
        const int s = i % NSLOT;
        q_wait(s, FILLED);
        if (q_fail) break;
        if (compression_kernel_wrapper(buf[s], BUF, bufout[s], 0, 0, 128, 0, s, in_d, out_d) != 1) { q_abort(); break; }
        q_set(s, LAUNCHED);
    }
    return NULL;
}
static void *t_cpu(void *a)
{
    (void)a;
    for (int i = 0; i < q_nbuf && !q_fail; i++) {
        const int s = i % NSLOT;
        q_wait(s, LAUNCHED);
        if (q_fail) break;
        if (!retire(s, &q_h, &q_bytes, q_hash)) { q_abort(); break; }
        q_set(s, FREE);
    }
    return NULL;
}
static double pass_threads(int nbuf, uint64_t *h, long long *bytes, int hash)
{
    pthread_t tp, tg, tc;
    q_nbuf = nbuf; q_hash = hash; q_fail = 0; q_h = 1469598103934665603ull; q_bytes = 0;
    for (int s = 0; s < NSLOT; s++) q_state[s] = FREE;
    const double t0 = now();
    pthread_create(&tp, NULL, t_producer, NULL); pthread_create(&tg, NULL, t_gpu, NULL); pthread_create(&tc, NULL, t_cpu, NULL);
    pthread_join(tp, NULL); pthread_join(tg, NULL); pthread_join(tc, NULL);
    const double t = now() - t0;
    *h = q_h; *bytes = q_bytes;
    return q_fail ? -1.0 : t;
}

int main(int argc, char **argv)
{
    const int nbuf = argc > 1 ? atoi(argv[1]) : 256;
    npool = argc > 2 ? atoi(argv[2]) : 16;
    if (nbuf < NSLOT || npool < 1) return 2;
    initGPU();
    in_d = initGPUmem(BUF); out_d = initGPUmem(BUF * 2);
    for (int s = 0; s < NSLOT; s++) { buf[s] = initCPUmem(BUF); bufout[s] = initCPUmem(BUF * 2); if (!buf[s] || !bufout[s]) return 2; }
    pool = (unsigned char **)malloc(sizeof(*pool) * (size_t)npool);
    for (int k = 0; k < npool; k++) { pool[k] = (unsigned char *)malloc((size_t)BUF); log_text(pool[k], BUF, 1000u + (uint32_t)k); }
    uint64_t h1, h2, h3, h4; long long b1, b2, b3, b4;
    if (pass(8, 1, &h1, &b1, 0) < 0 || pass(8, NSLOT, &h1, &b1, 0) < 0 || pass_threads(8, &h1, &b1, 0) < 0) { printf("FAILED warmup\n"); return 1; }
    const double tseq = pass(nbuf, 1, &h1, &b1, 0);
    const double tring = pass(nbuf, NSLOT, &h2, &b2, 0);
    const double tthr = pass_threads(nbuf, &h4, &b4, 0);
    /* bytes: the same passes once more with every packed byte hashed (the hash is not part of the pipeline's time) */
    const int nc = nbuf < 64 ? nbuf : 64;
    const double c1 = pass(nc, 1, &h1, &b1, 1), c2 = pass(nc, NSLOT, &h2, &b2, 1), c3 = pass(nc, 2, &h3, &b3, 1);
    const double c4 = pass_threads(nc, &h4, &b4, 1);
    const int same = c1 > 0 && c2 > 0 && c3 > 0 && c4 > 0 && h1 == h2 && b1 == b2 && h1 == h3 && b1 == b3 && h1 == h4 && b1 == b4;
    printf("nbuf=%d seq_GBps=%.4f ring_GBps=%.4f threads_GBps=%.4f seq_ms_per_buf=%.4f ring_ms_per_buf=%.4f threads_ms_per_buf=%.4f ratio=%.4f bytes_equal=%d hash=%016llx\n",
           nbuf, tseq > 0 ? (double)nbuf * BUF / tseq / 1e9 : 0.0, tring > 0 ? (double)nbuf * BUF / tring / 1e9 : 0.0,
           tthr > 0 ? (double)nbuf * BUF / tthr / 1e9 : 0.0, tseq * 1e3 / nbuf, tring * 1e3 / nbuf, tthr * 1e3 / nbuf,
           b1 ? (double)nc * BUF / (double)b1 : 0.0, same, (unsigned long long)h1);
    for (int s = 0; s < NSLOT; s++) { deleteCPUmem(buf[s]); deleteCPUmem(bufout[s]); }
    for (int k = 0; k < npool; k++) free(pool[k]);
    free(pool);
    deleteGPUmem(in_d); deleteGPUmem(out_d); deleteGPUStreams();
    return same && tseq > 0 && tring > 0 && tthr > 0 ? 0 : 1;
}
