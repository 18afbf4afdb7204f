/* l2sim.c -- development tool: predicted L2 miss traffic of the SpMM row gathers for a given
 * processing order / XCD partition (no GPU needed).  Model: each XCD walks its slice of the order
 * sequentially through a fully-associative LRU of `cap_lines` 128-byte lines (4 MiB L2 = 32768);
 * every gathered row touches row_lines consecutive lines (fp32 d=64: 2); the CSR stream and the
 * output row pass through the cache once as streaming lines.
 *
 *   gcc -O2 -o /tmp/l2sim tools/l2sim.c
 * called from tools/order_eval.py (binary files: indptr, indices, order, xcd_start[9]).           */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t *key; int32_t *prev, *next, *hnext; int32_t *bucket; int32_t head, tail, n, cap, nb; } lru_t;

static void lru_init(lru_t *c, int cap) {
    c->cap = cap; c->n = 0; c->head = c->tail = -1; c->nb = 1; while (c->nb < 2 * cap) c->nb <<= 1;
    c->key = malloc(sizeof(int64_t) * cap); c->prev = malloc(4 * cap); c->next = malloc(4 * cap);
    c->hnext = malloc(4 * cap); c->bucket = malloc(4 * c->nb);
    memset(c->bucket, 0xff, 4 * c->nb);
}
static inline uint32_t hsh(int64_t k, int nb) { return (uint32_t)((uint64_t)k * 0x9E3779B97F4A7C15ull >> 40) & (nb - 1); }
static void unlink_(lru_t *c, int i) {
    if (c->prev[i] >= 0) c->next[c->prev[i]] = c->next[i]; else c->head = c->next[i];
    if (c->next[i] >= 0) c->prev[c->next[i]] = c->prev[i]; else c->tail = c->prev[i];
}
static void push_front(lru_t *c, int i) {
    c->prev[i] = -1; c->next[i] = c->head;
    if (c->head >= 0) c->prev[c->head] = i; else c->tail = i;
    c->head = i;
}
/* returns 1 on hit */
static int lru_access(lru_t *c, int64_t k) {
    uint32_t b = hsh(k, c->nb);
    for (int i = c->bucket[b]; i >= 0; i = c->hnext[i])
        if (c->key[i] == k) { unlink_(c, i); push_front(c, i); return 1; }
    int i;
    if (c->n < c->cap) i = c->n++;
    else {
        i = c->tail; unlink_(c, i);
        uint32_t ob = hsh(c->key[i], c->nb);
        int *p = &c->bucket[ob];
        while (*p != i) p = &c->hnext[*p];
        *p = c->hnext[i];
    }
    c->key[i] = k; c->hnext[i] = c->bucket[b]; c->bucket[b] = i; push_front(c, i);
    return 0;
}

static void *slurp(const char *f, size_t *n) {
    FILE *fp = fopen(f, "rb"); if (!fp) { perror(f); exit(1); }
    fseek(fp, 0, SEEK_END); *n = ftell(fp); fseek(fp, 0, SEEK_SET);
    void *p = malloc(*n); if (fread(p, 1, *n, fp) != *n) exit(1); fclose(fp); return p;
}

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: l2sim indptr.i32 indices.i32 order.i32 xcd_start.i64 row_lines cap_lines\n"); return 2; }
    size_t n1, n2, n3, n4;
    int32_t *indptr = slurp(argv[1], &n1), *indices = slurp(argv[2], &n2), *order = slurp(argv[3], &n3);
    int64_t *xs = slurp(argv[4], &n4);
    const int row_lines = atoi(argv[5]), cap = atoi(argv[6]);
    const int W = argc > 7 ? atoi(argv[7]) : 1;       /* waves in flight per XCD (1 = sequential model) */
    const int RPW = argc > 8 ? atoi(argv[8]) : 4;     /* consecutive rows per wave */
    const int BATCH = 16;                             /* neighbours per gather batch */
    const int64_t N = n1 / 4 - 1;
    const int nx = (int)(n4 / 8) - 1;
    int64_t tot_miss = 0, tot_acc = 0, stream_id = (int64_t)1 << 40;
    for (int x = 0; x < nx; x++) {
        lru_t c; lru_init(&c, cap);
        int64_t miss = 0, acc = 0, nnz = 0;
        if (W <= 1) {
        for (int64_t p = xs[x]; p < xs[x + 1]; p++) {
            const int r = order[p];
            const int s = indptr[r], e = indptr[r + 1];
            for (int j = s; j < e; j++)
                for (int l = 0; l < row_lines; l++) { miss += !lru_access(&c, (int64_t)indices[j] * row_lines + l); acc++; }
            nnz += e - s;
            /* streaming pollution: CSR entries (8 B/nnz) and the output row */
            int64_t sl = ((int64_t)(e - s) * 8 + 127) / 128 + row_lines;
            for (int64_t t = 0; t < sl; t++) lru_access(&c, stream_id++);
        }
        } else {
            /* W waves in flight; a finished wave takes the next RPW rows (dispatch order); all waves
             * advance one gather batch per sweep */
            int64_t next = xs[x];
            int64_t *wp = malloc(sizeof(int64_t) * W), *wend = malloc(sizeof(int64_t) * W); int *wj = malloc(sizeof(int) * W);
            int live = 0;
            for (int w = 0; w < W; w++) { wp[w] = wend[w] = 0; wj[w] = 0; }
            for (;;) {
                live = 0;
                for (int w = 0; w < W; w++) {
                    if (wp[w] >= wend[w]) {
                        if (next >= xs[x + 1]) continue;
                        wp[w] = next; wend[w] = next + RPW < xs[x + 1] ? next + RPW : xs[x + 1]; next = wend[w];
                        wj[w] = indptr[order[wp[w]]];
                    }
                    live++;
                    const int r = order[wp[w]];
                    const int e = indptr[r + 1];
                    int j = wj[w];
                    const int je = j + BATCH < e ? j + BATCH : e;
                    for (; j < je; j++)
                        for (int l = 0; l < row_lines; l++) { miss += !lru_access(&c, (int64_t)indices[j] * row_lines + l); acc++; }
                    wj[w] = j;
                    if (j >= e) {
                        const int s0 = indptr[r];
                        nnz += e - s0;
                        int64_t sl = ((int64_t)(e - s0) * 8 + 127) / 128 + row_lines;
                        for (int64_t t = 0; t < sl; t++) lru_access(&c, stream_id++);
                        wp[w]++;
                        if (wp[w] < wend[w]) wj[w] = indptr[order[wp[w]]];
                    }
                }
                if (!live) break;
            }
            free(wp); free(wend); free(wj);
        }
        printf("  xcd %d: rows %lld nnz %lld gather lines %lld miss %lld (%.3f)\n", x, (long long)(xs[x + 1] - xs[x]),
               (long long)nnz, (long long)acc, (long long)miss, acc ? (double)miss / acc : 0.0);
        tot_miss += miss; tot_acc += acc;
        free(c.key); free(c.prev); free(c.next); free(c.hnext); free(c.bucket);
    }
    printf("TOTAL gather lines %lld miss %lld hit-rate %.4f miss MB %.1f\n", (long long)tot_acc, (long long)tot_miss,
           1.0 - (double)tot_miss / tot_acc, tot_miss * 128.0 / 1e6);
    (void)N;
    return 0;
}
