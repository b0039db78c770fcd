// Host-side check of the large-model Riccati step's static schedule (iterativelqr.jl_amd/csrc/ilqr_ric_schedule.hpp); built and
// run by tests/test_ric_schedule.py. Prints one line per tile count and exits non-zero on the first broken invariant.
#include <cstdio>
#include <cstdlib>
#include "ilqr_ric_schedule.hpp"

using namespace ilqr;

static void fail(int TN, const char* what, int a = -1, int b = -1) {
    std::fprintf(stderr, "TN=%d: %s (%d, %d)\n", TN, what, a, b);
    std::exit(1);
}

template <int TN>
static void check() {
    typedef RicSchedule<TN> RS;
    constexpr int NQ = RS::NQ;
    const auto& t = RS::tab;
    int uh[TN] = {}, qux[TN] = {}, quu = 0, T[NQ] = {}, Twin[NQ], Twave[NQ], qxx[NQ] = {}, p[NQ] = {};
    for (int q = 0; q < NQ; ++q) { Twin[q] = -1; Twave[q] = -1; }
    for (int w = 0; w < 4; ++w) {
        const int* lists[3] = {t.a[w], t.b[w], t.ct[w]};
        for (int win = 0; win < 3; ++win) {
            bool ended = false;
            for (int i = 0; i < RS::MAXL; ++i) {
                const int task = lists[win][i];
                if (task < 0) { ended = true; continue; }
                if (ended) fail(TN, "a task behind the end of its list", w, win);
                const int kind = task & 0xe0, idx = task & 0x1f;
                if (kind == RIC_UH) { if (win != 0 || idx >= TN) fail(TN, "ûx tile outside window A", w, idx); uh[idx]++; }
                else if (kind == RIC_T) { if (idx >= NQ) fail(TN, "T tile index", w, idx); T[idx]++; Twin[idx] = win; Twave[idx] = w; }
                else if (kind == RIC_QUX) { if (win != 1 || idx >= TN) fail(TN, "Qux tile outside window B", w, idx); qux[idx]++; }
                else if (kind == RIC_QUU) { if (win != 1) fail(TN, "Quu outside window B", w, win); quu++; }
                else fail(TN, "unknown task kind", w, task);
                if (win == 2 && w == 0) fail(TN, "wave 0 runs the chain in window C: no tile there", w, task);
            }
        }
        for (int s = 0; s < RS::SLOTS; ++s) {
            const int e = t.qxx[w][s];
            if (e >= 0) {
                if (w == 0) fail(TN, "wave 0 runs the chain in window C: no Qxx tile there", s, e);
                const int q = e & ~RS::RIC_WAIT_T;
                if (q >= NQ) fail(TN, "Qxx tile index", w, q);
                qxx[q]++;
            }
            const int pq = t.p[w][s];
            if (pq >= 0) { if (pq >= NQ) fail(TN, "P tile index", w, pq); p[pq]++; }
        }
    }
    for (int c = 0; c < TN; ++c) if (uh[c] != 1 || qux[c] != 1) fail(TN, "ûx / Qux tile not formed exactly once", c, uh[c] * 10 + qux[c]);
    if (quu != 1) fail(TN, "Quu not formed exactly once", quu);
    for (int q = 0; q < NQ; ++q) if (T[q] != 1 || qxx[q] != 1 || p[q] != 1) fail(TN, "T / Qxx / P tile not formed exactly once", q, T[q] * 100 + qxx[q] * 10 + p[q]);
    // Qxx(a, c) = sum_k T(a, k) fx(k, c): every T(a, k) formed in window C must be the tile's own wave's (its T list runs first) or
    // the tile must wait for the flags of the other waves that form T there
    for (int w = 1; w < 4; ++w)
        for (int s = 0; s < RS::SLOTS; ++s) {
            const int e = t.qxx[w][s];
            if (e < 0) continue;
            const int q = e & ~RS::RIC_WAIT_T, a = q / TN;
            for (int k = 0; k < TN; ++k) {
                const int tq = a * TN + k;
                if (Twin[tq] == 2 && Twave[tq] != w && !(e & RS::RIC_WAIT_T)) fail(TN, "Qxx tile reads a T tile another wave forms in the same window, without waiting", q, tq);
            }
        }
    int per_wave_c[4] = {0, 0, 0, 0};
    for (int w = 1; w < 4; ++w) {
        for (int i = 0; i < RS::MAXL; ++i) per_wave_c[w] += t.ct[w][i] >= 0;
        for (int s = 0; s < RS::SLOTS; ++s) per_wave_c[w] += t.qxx[w][s] >= 0;
    }
    std::printf("TN=%d: %d T, %d Qxx, %d P tiles, %d + 1 Qux/Quu; window C tiles per wave %d %d %d\n", TN, NQ, NQ, NQ, TN, per_wave_c[1], per_wave_c[2], per_wave_c[3]);
}

// ---- the role rule of DESIGN.md §3.0 (role_mask): on every placement of up to four two-wave workgroups on a CU's four SIMDs the
// chosen assignment covers as many distinct SIMDs as ANY assignment can, with as few swaps as that allows, and every workgroup
// (whatever its own index) reads the same answer off the same table; the two placements seen on MI355X come out as expected.
static int covered_by(const int* e, int n, int mask) {
    int used = 0;
    for (int i = 0; i < n; ++i) used |= 1 << (((mask >> i) & 1) ? (e[i] >> 2) & 3 : e[i] & 3);
    int c = 0;
    for (int b = 0; b < 4; ++b) c += (used >> b) & 1;
    return c;
}
static void check_roles() {
    int cases = 0;
    for (int n = 1; n <= 4; ++n) {
        int total = 1;
        for (int i = 0; i < n; ++i) total *= 12;                       // ordered pairs of distinct SIMDs per workgroup
        for (int code = 0; code < total; ++code) {
            int e[4], c = code;
            for (int i = 0; i < n; ++i) {
                const int pr = c % 12; c /= 12;
                const int s0 = pr / 3, o = pr % 3, s1 = o >= s0 ? o + 1 : o;
                e[i] = role_entry(s0, s1);
            }
            const int m = role_mask(e, n);
            int best_cov = 0, best_swaps = 99;
            for (int mask = 0; mask < (1 << n); ++mask) {
                const int cov = covered_by(e, n, mask), sw = __builtin_popcount(mask);
                if (cov > best_cov || (cov == best_cov && sw < best_swaps)) { best_cov = cov; best_swaps = sw; }
            }
            if (covered_by(e, n, m) != best_cov || __builtin_popcount(m) != best_swaps) fail(0, "role_mask is not optimal", n, code);
            ++cases;
        }
    }
    // the ordinary CU of a full chip: (2,1) (3,0) (0,2) (1,3) — nobody swaps; the one in sixteen with two first waves on SIMD 0:
    // (0,2) (2,1) (1,3) (0,3) — the last workgroup swaps
    const int normal[4] = {role_entry(2, 1), role_entry(3, 0), role_entry(0, 2), role_entry(1, 3)};
    const int odd[4] = {role_entry(0, 2), role_entry(2, 1), role_entry(1, 3), role_entry(0, 3)};
    if (role_mask(normal, 4) != 0) fail(0, "ordinary placement: no swap expected");
    if (role_mask(odd, 4) != 8) fail(0, "two critical waves on SIMD 0: the fourth workgroup swaps", role_mask(odd, 4));
    // entries that have not arrived are ignored
    const int partial[4] = {role_entry(0, 2), 0, role_entry(0, 3), 0};
    if (role_mask(partial, 3) != 1) fail(0, "absent entries (one swap covers both SIMDs: the lowest such mask)", role_mask(partial, 3));
    std::printf("roles: %d placements checked\n", cases);
}

int main() {
    check<1>(); check<2>(); check<3>(); check<4>();
    check_roles();
    return 0;
}
