// lgcn_host.cpp -- host half of the C ABI (include/lgcn_hip.h): the BPR triplet
// samplers, the epoch shuffle and the normalised-adjacency builder.
//
// These replace, bit for bit:
//   sources/sampling.cpp:22-106   (pybind11 module `sampling`, glibc rand() stream)
//   utils.py:84-110               (UniformSample_original_python, numpy legacy stream)
//   utils.py:142-151              (utils.shuffle -> np.random.shuffle)
//   dataloader.py:133-136,218-234 (UserItemNet CSR, A_hat = D^-1/2 A D^-1/2)
// Generators are restated from their published algorithms (glibc random_r.c TYPE_3;
// MT19937 + numpy's masked-rejection bounded integers) and own their state, so the
// stream does not depend on who else calls rand() in the process.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "lgcn_hip.h"

namespace {

thread_local std::string g_error;

// glibc TYPE_3 additive feedback generator: x[n] = x[n-3] + x[n-31] mod 2^32, out = x>>1
class GlibcRand {
public:
    GlibcRand() { seed(1); }
    void seed(uint32_t s) {
        if (s == 0) s = 1;
        int32_t w = (int32_t)s;
        ring_[0] = (uint32_t)w;
        for (int i = 1; i < kDeg; i++) {          // Park-Miller minimal standard via Schrage
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            ring_[i] = (uint32_t)w;
        }
        front_ = kSep; rear_ = 0;
        for (int i = 0; i < 10 * kDeg; i++) next();
    }
    inline int next() {
        const uint32_t v = ring_[front_] += ring_[rear_];
        if (++front_ == kDeg) front_ = 0;
        if (++rear_ == kDeg) rear_ = 0;
        return (int)(v >> 1);
    }
    inline int below(int end) { return next() % end; }     // sampling.cpp:22-25

    // The generator is the linear recurrence x[n] = x[n-3] + x[n-31] over Z/2^32; its state is the
    // history h[j] = x[n-31+j], j = 0..30 (oldest first).  One step maps h to (h[1..30], h[0]+h[28]), a
    // 31 x 31 matrix M over Z/2^32 -- so n steps are M^n, computed by squaring: the stream can be
    // entered at any position without walking there (the GPU sampler expands blocks of it in
    // parallel from their start histories, and the host generator is moved past what the GPU used).
    void history(uint32_t h[31]) const { for (int j = 0; j < kDeg; j++) h[j] = ring_[(front_ + j) % kDeg]; }
    void set_history(const uint32_t h[31]) { for (int j = 0; j < kDeg; j++) ring_[j] = h[j]; front_ = 0; rear_ = kDeg - kSep; }
    struct Mat { uint32_t a[31][31]; };
    static Mat step_matrix() {
        Mat m; std::memset(&m, 0, sizeof m);
        for (int j = 0; j < kDeg - 1; j++) m.a[j][j + 1] = 1;
        m.a[kDeg - 1][0] = 1; m.a[kDeg - 1][kDeg - kSep] = 1;
        return m;
    }
    static Mat mul(const Mat &x, const Mat &y) {
        Mat r; std::memset(&r, 0, sizeof r);
        for (int i = 0; i < kDeg; i++)
            for (int k = 0; k < kDeg; k++) {
                const uint32_t xv = x.a[i][k];
                if (!xv) continue;
                for (int j = 0; j < kDeg; j++) r.a[i][j] += xv * y.a[k][j];
            }
        return r;
    }
    static Mat power(uint64_t n) {            // M^n
        Mat r; std::memset(&r, 0, sizeof r);
        for (int i = 0; i < kDeg; i++) r.a[i][i] = 1;
        Mat b = step_matrix();
        for (; n; n >>= 1) { if (n & 1) r = mul(b, r); b = mul(b, b); }
        return r;
    }
    static void apply(const Mat &m, const uint32_t in[31], uint32_t out[31]) {
        for (int i = 0; i < kDeg; i++) { uint32_t acc = 0; for (int j = 0; j < kDeg; j++) acc += m.a[i][j] * in[j]; out[i] = acc; }
    }
    void jump(uint64_t n) {                   // as if next() had been called n times
        uint32_t h[31], o[31];
        history(h);
        const Mat m = power(n);
        apply(m, h, o);
        set_history(o);
    }

private:
    static constexpr int kDeg = 31, kSep = 3;
    uint32_t ring_[kDeg];
    int front_, rear_;
};

// numpy legacy RandomState bit stream
class Mt19937 {
public:
    Mt19937() { seed(5489u); }
    void seed(uint32_t s) {
        mt_[0] = s;
        for (uint32_t i = 1; i < kN; i++) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + i;
        idx_ = kN;
    }
    inline uint32_t next() {
        if (idx_ >= kN) refill();
        return out_[idx_++];
    }
    // uniform integer in [0, top] the way numpy's legacy bounded integers draw it
    inline uint32_t upto(uint32_t top) {
        if (top == 0) return 0;
        if (top == 0xffffffffu) return next();
        uint32_t mask = top;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next() & mask; } while (v > top);
        return v;
    }

    // the whole state, for the device form of the shuffle (lgcn_shuffle.hip): key[624] + read position
    void get_state(uint32_t *key, uint32_t *pos) const { std::memcpy(key, mt_, sizeof mt_); *pos = idx_; }
    void set_state(const uint32_t *key, uint32_t pos) { std::memcpy(mt_, key, sizeof mt_); idx_ = pos > kN ? kN : pos; temper(); }

private:
    static constexpr uint32_t kN = 624, kM = 397;
    static inline uint32_t twist(uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ (0u - (y & 1u) & 0x9908b0dfu);
    }
    // the three index ranges of the twist without modulo arithmetic (the first two vectorise: word i reads words i+1 and
    // i+397 / i-227, all still old or all already new), then the whole block is tempered at once
    void refill() {
        uint32_t i = 0;
        for (; i < kN - kM; i++) mt_[i] = twist(mt_[i], mt_[i + 1], mt_[i + kM]);
        for (; i < kN - 1; i++) mt_[i] = twist(mt_[i], mt_[i + 1], mt_[i + kM - kN]);
        mt_[kN - 1] = twist(mt_[kN - 1], mt_[0], mt_[kM - 1]);
        temper();
        idx_ = 0;
    }
    void temper() {
        for (uint32_t i = 0; i < kN; i++) {
            uint32_t y = mt_[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out_[i] = y;
        }
    }
    uint32_t mt_[kN];
    uint32_t out_[kN];        // the tempered outputs of the current block
    uint32_t idx_;
};

GlibcRand g_rand;
Mt19937 g_np;

// std::find over a user's positives (sampling.cpp:48-49, utils.py:105).  The rows
// are sorted ascending (scipy canonical CSR), so membership is a binary search.
// Branch-free: the answer is "no" almost every time and the probes of a taken / not-taken search are coin flips to the
// predictor (measured on the python-mode Gowalla epoch: 60 -> 4x ms with the prefetches in lgcn_sample_python).
inline bool is_positive(const int32_t *row, int deg, int item) {
    if (deg <= 0) return false;
    const int32_t *base = row;
    int n = deg;
    while (n > 1) {                                   // invariant: the last element <= item, if any, lies in [base, base + n)
        const int half = n >> 1;
        base = base[half] <= item ? base + half : base;      // (compiles to a conditional move)
        n -= half;
    }
    return *base == item;
}

inline bool rows_sorted(const int64_t *indptr, const int32_t *indices, int n) {
    for (int u = 0; u < n; u++)
        for (int64_t p = indptr[u] + 1; p < indptr[u + 1]; p++)
            if (indices[p - 1] >= indices[p]) return false;
    return true;
}

}  // namespace

extern "C" {

void lgcn_set_error(const char *msg) { g_error = msg ? msg : ""; }
const char *lgcn_last_error(void) { return g_error.c_str(); }
int lgcn_abi_version(void) { return LGCN_ABI_VERSION; }

void lgcn_sampling_seed(unsigned int seed) { g_rand.seed(seed); }
int lgcn_sampling_randint(int end) {
    if (end <= 0) { lgcn_set_error("randint: end must be positive"); return -1; }
    return g_rand.below(end);
}

int lgcn_sample_negative(int user_num, int item_num, int64_t train_num, const int64_t *indptr,
                         const int32_t *indices, int neg_num, int32_t *S_out) {
    if (user_num <= 0 || item_num <= 0 || train_num < 0 || neg_num < 0 || !indptr || !indices || !S_out) {
        lgcn_set_error("sample_negative: invalid argument");
        return 3;
    }
    if (!rows_sorted(indptr, indices, user_num)) {
        lgcn_set_error("sample_negative: positives must be sorted ascending per user (CSR canonical form)");
        return 3;
    }
    const int per_user = (int)(train_num / user_num);
    const int width = neg_num + 2;
    if (per_user > 0)
        for (int u = 0; u < user_num; u++)
            if (indptr[u + 1] == indptr[u]) {
                lgcn_set_error("sample_negative: a user has no training positives (the reference divides by zero "
                               "here); use the python-mode sampler for such datasets");
                return 2;
            }
    int32_t *o = S_out;
    for (int u = 0; u < user_num; u++) {
        const int32_t *row = indices + indptr[u];
        const int deg = (int)(indptr[u + 1] - indptr[u]);
        if (deg >= item_num && neg_num > 0 && per_user > 0) {
            lgcn_set_error("sample_negative: a user is positive on every item (rejection loop would not end)");
            return 2;
        }
        for (int k = 0; k < per_user; k++, o += width) {
            o[0] = u;
            o[1] = row[g_rand.below(deg)];
            for (int j = 2; j < width; j++) {
                int cand;
                do { cand = g_rand.below(item_num); } while (is_positive(row, deg, cand));
                o[j] = cand;
            }
        }
    }
    return 0;
}

int lgcn_sample_negative_by_user(const int32_t *users, int n_listed, int item_num, const int64_t *indptr,
                                 const int32_t *indices, int neg_num, int32_t *S_out) {
    if (n_listed < 0 || item_num <= 0 || neg_num < 0 || !users || !indptr || !indices || !S_out) {
        lgcn_set_error("sample_negative_ByUser: invalid argument");
        return 3;
    }
    const int width = neg_num + 2;
    for (int i = 0; i < n_listed; i++) {
        const int u = users[i];
        const int32_t *row = indices + indptr[u];
        const int deg = (int)(indptr[u + 1] - indptr[u]);
        if (deg == 0 || (deg >= item_num && neg_num > 0)) {
            lgcn_set_error("sample_negative_ByUser: user without positives / without negatives");
            return 2;
        }
        if (!std::is_sorted(row, row + deg)) {
            lgcn_set_error("sample_negative_ByUser: positives must be sorted ascending");
            return 3;
        }
        int32_t *o = S_out + (int64_t)i * width;
        o[0] = u;
        o[1] = row[g_rand.below(deg)];
        for (int j = 2; j < width; j++) {
            int cand;
            do { cand = g_rand.below(item_num); } while (is_positive(row, deg, cand));
            o[j] = cand;
        }
    }
    return 0;
}

// ---- glibc stream access for the GPU sampler (csrc/lgcn_sampler.hip)
// start histories of `nblocks` consecutive blocks of `block_len` draws, beginning at the generator's
// current position: out[b*31 + j]
void lgcn_glibc_block_histories(int64_t nblocks, int64_t block_len, uint32_t *out) {
    uint32_t h[31], o[31];
    g_rand.history(h);
    const GlibcRand::Mat m = GlibcRand::power((uint64_t)block_len);
    for (int64_t b = 0; b < nblocks; b++) {
        std::memcpy(out + b * 31, h, sizeof h);
        GlibcRand::apply(m, h, o);
        std::memcpy(h, o, sizeof h);
    }
}
void lgcn_glibc_advance(uint64_t n) { g_rand.jump(n); }

void lgcn_np_seed(uint32_t seed) { g_np.seed(seed); }
void lgcn_np_get_state(uint32_t *key624, uint32_t *pos) { g_np.get_state(key624, pos); }
void lgcn_np_set_state(const uint32_t *key624, uint32_t pos) { g_np.set_state(key624, pos); }

int64_t lgcn_sample_python(int n_users, int m_items, int64_t train_num, const int64_t *indptr,
                           const int32_t *indices, int64_t *S_out) {
    if (n_users <= 0 || m_items <= 0 || train_num < 0 || !indptr || !indices || !S_out) {
        lgcn_set_error("sample_python: invalid argument");
        return -1;
    }
    if (!rows_sorted(indptr, indices, n_users)) {
        lgcn_set_error("sample_python: positives must be sorted ascending per user");
        return -1;
    }
    // users = np.random.randint(0, n_users, trainDataSize) is drawn in full first (utils.py:93)
    std::vector<int32_t> drawn((size_t)train_num);
    for (int64_t t = 0; t < train_num; t++) drawn[(size_t)t] = (int32_t)g_np.upto((uint32_t)n_users - 1);
    int64_t rows = 0;
    for (int64_t t = 0; t < train_num; t++) {
        // the users are known ahead (drawn in full above): their row bounds and the head of their rows are fetched early --
        // the loop is otherwise one dependent cache miss per user behind the serial generator
        if (t + 16 < train_num) __builtin_prefetch(indptr + drawn[(size_t)t + 16]);
        if (t + 8 < train_num) __builtin_prefetch(indices + indptr[drawn[(size_t)t + 8]]);
        const int u = drawn[(size_t)t];
        const int32_t *row = indices + indptr[u];
        const int deg = (int)(indptr[u + 1] - indptr[u]);
        if (deg == 0) continue;                                       // utils.py:99-100
        if (deg >= m_items) { lgcn_set_error("sample_python: user positive on every item"); return -1; }
        const int32_t positive = row[g_np.upto((uint32_t)deg - 1)];  // np.random.choice(posForUser)
        int32_t cand;
        do { cand = (int32_t)g_np.upto((uint32_t)m_items - 1); } while (is_positive(row, deg, cand));
        int64_t *o = S_out + rows * 3;
        o[0] = u; o[1] = positive; o[2] = cand;
        rows++;
    }
    return rows;
}

int lgcn_np_shuffle_perm(int64_t n, int64_t *perm) {
    if (n < 0 || (n > 0 && !perm) || n > 0xffffffffLL) { lgcn_set_error("shuffle_perm: invalid argument"); return 3; }
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    for (int64_t i = n - 1; i >= 1; i--) {          // Fisher-Yates from the top, j in [0,i]
        const int64_t j = (int64_t)g_np.upto((uint32_t)i);
        std::swap(perm[i], perm[j]);
    }
    return 0;
}

int lgcn_build_user_item_csr(int n_users, int m_items, int64_t n_inter, const int64_t *tu, const int64_t *ti,
                             int64_t *indptr, int32_t *indices, float *vals, int64_t *nnz_out) {
    if (n_users <= 0 || m_items <= 0 || n_inter < 0 || !tu || !ti || !indptr || !nnz_out) {
        lgcn_set_error("build_user_item_csr: invalid argument");
        return 3;
    }
    std::vector<int64_t> start((size_t)n_users + 1, 0);
    for (int64_t e = 0; e < n_inter; e++) {
        if (tu[e] < 0 || tu[e] >= n_users || ti[e] < 0 || ti[e] >= m_items) {
            lgcn_set_error("build_user_item_csr: id out of range");
            return 3;
        }
        start[(size_t)tu[e] + 1]++;
    }
    for (int u = 0; u < n_users; u++) start[(size_t)u + 1] += start[(size_t)u];
    std::vector<int32_t> cols((size_t)n_inter);
    {
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (int64_t e = 0; e < n_inter; e++) cols[(size_t)cur[(size_t)tu[e]]++] = (int32_t)ti[e];
    }
    int64_t nnz = 0;
    indptr[0] = 0;
    for (int u = 0; u < n_users; u++) {
        int32_t *b = cols.data() + start[(size_t)u], *e = cols.data() + start[(size_t)u + 1];
        std::sort(b, e);
        for (int32_t *p = b; p < e;) {
            int32_t *q = p;
            while (q < e && *q == *p) q++;
            if (indices) { indices[nnz] = *p; vals[nnz] = (float)(q - p); }
            nnz++;
            p = q;
        }
        indptr[u + 1] = nnz;
    }
    *nnz_out = nnz;
    return 0;
}

int lgcn_adj_rowsum(int n_users, int m_items, const int64_t *rp, const int32_t *ri, const float *rv, float *rowsum) {
    if (n_users <= 0 || m_items <= 0 || !rp || !ri || !rv || !rowsum) { lgcn_set_error("adj_rowsum: invalid argument"); return 3; }
    std::fill(rowsum, rowsum + (size_t)n_users + (size_t)m_items, 0.0f);
    for (int u = 0; u < n_users; u++) {
        float s = 0.0f;
        for (int64_t p = rp[u]; p < rp[u + 1]; p++) {
            s += rv[p];
            rowsum[(size_t)n_users + (size_t)ri[p]] += rv[p];
        }
        rowsum[u] = s;
    }
    return 0;
}

int lgcn_build_norm_adj(int n_users, int m_items, const int64_t *rp, const int32_t *ri, const float *rv,
                        const float *d_inv, int32_t *indptr, int32_t *indices, float *data) {
    if (n_users <= 0 || m_items <= 0 || !rp || !ri || !rv || !d_inv || !indptr || !indices || !data) {
        lgcn_set_error("build_norm_adj: invalid argument");
        return 3;
    }
    const int64_t E = rp[n_users];
    if (2 * E > 0x7fffffffLL) { lgcn_set_error("build_norm_adj: nnz exceeds int32 indexing"); return 3; }
    // upper block rows (users): the user's item list shifted by n_users
    std::vector<int32_t> item_deg((size_t)m_items, 0);
    indptr[0] = 0;
    for (int u = 0; u < n_users; u++) {
        indptr[u + 1] = (int32_t)rp[u + 1];
        const float du = d_inv[u];
        for (int64_t p = rp[u]; p < rp[u + 1]; p++) {
            const int i = ri[p];
            item_deg[(size_t)i]++;
            indices[p] = n_users + i;
            data[p] = (du * rv[p]) * d_inv[(size_t)n_users + (size_t)i];
        }
    }
    // lower block rows (items) = R^T: counting transpose; users arrive ascending, so
    // columns come out sorted
    int32_t *lp = indptr + n_users;
    for (int i = 0; i < m_items; i++) lp[i + 1] = lp[i] + item_deg[(size_t)i];
    std::vector<int32_t> fill(lp, lp + m_items);
    for (int u = 0; u < n_users; u++) {
        const float du = d_inv[u];
        for (int64_t p = rp[u]; p < rp[u + 1]; p++) {
            const int i = ri[p];
            const int32_t q = fill[(size_t)i]++;
            indices[q] = u;
            data[q] = (d_inv[(size_t)n_users + (size_t)i] * rv[p]) * du;
        }
    }
    return 0;
}

}  // extern "C"
