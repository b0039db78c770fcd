// Static schedule of the large-model Riccati step (ilqr_device_large.hpp: backward_pass_large_fn): which wave forms which 16x16
// tile in which of the step's four windows. Plain constexpr C++ (no HIP), so that the host-side test can check its invariants
// (tests/test_ric_schedule.py: every tile exactly once, every Qxx tile behind the T tiles it reads, no list overflow).
#pragma once

namespace ilqr {

enum { RIC_UH = 1 << 5, RIC_T = 2 << 5, RIC_QUX = 3 << 5, RIC_QUU = 4 << 5, RIC_END = 0xff };   // task byte: kind << 5 | tile index
template <int TN>
struct RicSchedule {
    static constexpr int NQ = TN * TN, MAXL = 8, SLOTS = (NQ + 2) / 3, RIC_WAIT_T = 0x40;       // at most MAXL tasks per wave and window
    struct Tab {
        int a[4][MAXL], b[4][MAXL], ct[4][MAXL], qxx[4][SLOTS], p[4][SLOTS];
    };
    static constexpr Tab make() {
        Tab t{};
        int na[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, nc[4] = {0, 0, 0, 0}, nq[4] = {0, 0, 0, 0}, np[4] = {0, 0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            for (int i = 0; i < MAXL; ++i) { t.a[w][i] = -1; t.b[w][i] = -1; t.ct[w][i] = -1; }
            for (int i = 0; i < SLOTS; ++i) { t.qxx[w][i] = -1; t.p[w][i] = -1; }
        }
        if (TN == 2) {
            // Round 5: THREE windows per step. Window A is gone: every wave forms ONE tile of P in window D and, straight from that
            // tile's registers (the D layout of a tile is the B layout of the next MFMA), its share of the next step's ûx = fuᵀP′ —
            // the partial sum over the tile's 16 rows of P′; the consumers (Qux, Quu in window B) add the two partial sums of a
            // column block when they read their A fragments. One LDS round trip, one barrier and four of ûx's eight dependent MFMAs
            // are off the critical path. The two matrix-vector products of the step ride in the PADDING of those tiles: p′ᵀ is row
            // nu of ûx (nu < 16), so row nu of Qux = ûx fx is (fxᵀp′)ᵀ and row nu of Quu = ûx fu is (fuᵀp′)ᵀ — wave 3, which used to
            // form them in window B, takes a tile of T there instead. The list `a` is the PROLOGUE's: ûx of the first step from
            // P[H] in LDS.
            //   B: Qux, Quu (with Qx, Qu) | wave 3: T(0,0)   C: chain | T(0,1), T(1,0), T(1,1), Qxx   D (+A): P, partial ûx
            t.a[0][na[0]++] = RIC_UH + 0; t.a[1][na[1]++] = RIC_UH + 1;
            t.b[0][nb[0]++] = RIC_QUX + 0; t.b[1][nb[1]++] = RIC_QUX + 1; t.b[2][nb[2]++] = RIC_QUU; t.b[3][nb[3]++] = RIC_T + 0;
            t.ct[1][nc[1]++] = RIC_T + 1; t.ct[2][nc[2]++] = RIC_T + 2; t.ct[3][nc[3]++] = RIC_T + 3;
            // Qxx(0,.) reads T(0,0) (window B) and T(0,1) (its own wave's); Qxx(1,.) needs T(1,0) and T(1,1), formed in this same
            // window by waves 2 and 3: those tiles wait for the flags of the OTHER waves that form T here (RIC_WAIT_T)
            t.qxx[1][nq[1]++] = 0; t.qxx[1][nq[1]++] = 1; t.qxx[2][nq[2]++] = 2 | RIC_WAIT_T; t.qxx[3][nq[3]++] = 3 | RIC_WAIT_T;
            t.p[0][np[0]++] = 3; t.p[1][np[1]++] = 0; t.p[2][np[2]++] = 1; t.p[3][np[3]++] = 2;
        } else {
            for (int c = 0; c < TN; ++c) { const int w = c % 2; t.a[w][na[w]++] = RIC_UH + c; }
            for (int v = 2; v < 4; ++v) {
                const int cnt = (NQ - (v - 2) + 1) / 2, first = (cnt + 1) / 2;
                for (int k = 0; k < cnt; ++k) {
                    const int q = (v - 2) + 2 * k;
                    if (k < first) t.a[v][na[v]++] = RIC_T + q; else t.b[v][nb[v]++] = RIC_T + q;
                }
            }
            for (int i = 0; i <= TN; ++i) { const int w = i % 2; t.b[w][nb[w]++] = i < TN ? RIC_QUX + i : RIC_QUU; }
            for (int q = 0; q < NQ; ++q) { const int w = 1 + (q + 1) % 3; t.qxx[w][nq[w]++] = q; }
            for (int q = 0; q < NQ; ++q) { const int w = (q + 1) % 4; t.p[w][np[w]++] = q; }      // wave 0 last: it also stores p, Lx, Lu
        }
        return t;
    }
    static constexpr Tab tab = make();
    static_assert(NQ <= 16, "tile index fits five bits (TN <= 4)");
};

// ---- which wave of each workgroup of a CU is its critical one (ilqr_device.hpp: pick_roles). Plain constexpr C++
// like the schedule above, so that the rule is checked on the host (tests/test_ric_schedule.py). Entry i of a CU's table:
// bit 4 = present, bits 0-1 the SIMD of the workgroup's first wave, bits 2-3 that of its second. Bit i of the result = workgroup i
// swaps its roles (its SECOND wave is the critical one). Of all assignments: the most distinct SIMDs under critical waves, then the
// fewest swaps, then the lowest mask — one answer for every workgroup that evaluates it on the same table.
constexpr int role_entry(int s0, int s1) { return 16 | (s0 & 3) | ((s1 & 3) << 2); }
constexpr int role_mask(const int* e, int n) {
    int best = 0, best_score = -1;
    for (int mask = 0; mask < (1 << n); ++mask) {
        int used = 0, swaps = 0;
        for (int i = 0; i < n; ++i) {
            if (!(e[i] & 16)) continue;
            const int sw = (mask >> i) & 1;
            used |= 1 << (sw ? (e[i] >> 2) & 3 : e[i] & 3);
            swaps += sw;
        }
        int covered = 0;
        for (int b = 0; b < 4; ++b) covered += (used >> b) & 1;
        const int score = 16 * covered - swaps;
        if (score > best_score) { best_score = score; best = mask; }
    }
    return best;
}

}  // namespace ilqr
