// mfp_march.h — shared pieces of the wave-marching kernels (mfp_march.hip: hydro, mfp_march_field.hip: field).
#pragma once
#include <cstdlib>

#include "mfp_device.h"

namespace CMFP_NS {
using namespace cmfp;

// value of the neighbouring lane of the wavefront (wave_shr:1 / wave_shl:1); every lane must execute these.
// bound_ctrl: the lane without a neighbour gets 0 — what it got before from the `old` operand 0, but now the
// destination does not depend on `old` and the compiler no longer zeroes it in front of every shift (two
// v_mov per shifted double: 5 % of the instructions of the hydro marching kernel)
CMFP_DEV double lane_left(double v) {  // lane l <- lane l - 1
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
CMFP_DEV double lane_right(double v) {  // lane l <- lane l + 1
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct MarchBC {
    int touch[2][2];     // [d][side]: physical (non-periodic) domain face on this block
    int rule[2][2][10];  // FILL_SKIP / FILL_EVEN / FILL_ODD per primitive comp
};
// Cells a launch updates: up to four rectangles (the whole block; its interior; the frame around
// the interior as four rectangles), each cut into strips of `mw` columns and segments of `seg` rows.
struct MarchRect {
    int ilo, ihi, jlo, jhi;
    int ns, nseg;  // strips, segments
    int first;     // index of the rectangle's first wave
};
struct MarchSel {
    int nrect, nwaves, seg;
    int wrap;  // bit d set: direction d is periodic on this (unsplit) block and the kernel wraps its loads itself
               // instead of reading ghost cells (then no ghost fill is launched for that direction)
    MarchRect r[4];
};
// `part` arguments of the 2-D launchers: low byte 0 / 1 / 2 (all cells / interior / frame), bit 8 the LDS-tiled
// kernel only, bits 9 and 10 the in-kernel periodic wrap of x and y (marching kernels only)
static inline int part_wrap(int part) { return (part >> 9) & 3; }
CMFP_DEV int wrap_index(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

static inline MarchBC make_march_bc(const StatePhys& sp, const int touches[3][2]) {
    MarchBC b;
    for (int d = 0; d < 2; ++d)
        for (int s = 0; s < 2; ++s) {
            b.touch[d][s] = touches[d][s];
            for (int n = 0; n < 10; ++n) {
                const int code = (n < sp.nc) ? sp.bc[n][d][s] : -666;  // MFP_eulerian.cpp:796-835
                b.rule[d][s][n] = (code == 2 || code == 4 || code == 1) ? FILL_EVEN : (code == -1 ? FILL_ODD : FILL_SKIP);
            }
        }
    return b;
}

// wave slots of the device at two waves per SIMD (the kernels' register budget)
static inline int wave_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0;
        hipDeviceProp_t p;
        slots = 2048;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            slots = p.multiProcessorCount * 8;
    }
    return slots;
}

// Rectangles, strips and segments of one launch.  ring = 1 + stencil half width (the ghost reach of
// the update), mw = columns a strip updates, mult = waves per (strip, segment) (field: one per
// component group).  part 0 all cells, 1 the interior (stencil footprint inside valid cells), 2 the
// frame around it.  Returns false when the launch is empty.
// the cells of the interior launch (part 1): stencil footprint (+- ring) inside the valid cells
static inline bool march_interior(const Geo& g, int ring, IBox& box) {
    box.lo[0] = ring, box.hi[0] = g.n[0] - 1 - ring;
    box.lo[1] = ring, box.hi[1] = g.n[1] - 1 - ring;
    box.lo[2] = box.hi[2] = 0;
    return box.hi[0] >= box.lo[0] && box.hi[1] >= box.lo[1];
}
static inline bool march_select(const Geo& g, int part, int ring, int mw, int mult, MarchSel& sel) {
    const int n0 = g.n[0], n1 = g.n[1];
    IBox inner;
    const bool has_int = march_interior(g, ring, inner);
    sel.nrect = 0;
    auto add = [&](int ilo, int ihi, int jlo, int jhi) {
        if (ihi < ilo || jhi < jlo) return;
        MarchRect& r = sel.r[sel.nrect++];
        r.ilo = ilo, r.ihi = ihi, r.jlo = jlo, r.jhi = jhi;
        r.ns = (ihi - ilo + mw) / mw;
    };
    // frame widths = the ghost reach.  They must not grow: the field's launch applies the source epilogue to
    // its interior tiles on the second stream and reads the species there, so every cell of a field interior
    // tile (>= 14 cells from the faces) has to lie in the hydro interior launched before it on that stream.
    // (A strip-wide frame was 2.5 % faster and wrong for exactly that reason.)
    const int fx = ring, fy = ring;
    if (part == 0 || (part == 2 && !has_int)) {
        add(0, n0 - 1, 0, n1 - 1);
    } else if (part == 1) {
        add(inner.lo[0], inner.hi[0], inner.lo[1], inner.hi[1]);
    } else {
        add(0, n0 - 1, 0, fy - 1);
        add(0, n0 - 1, n1 - fy, n1 - 1);
        add(0, fx - 1, fy, n1 - 1 - fy);
        add(n0 - fx, n0 - 1, fy, n1 - 1 - fy);
    }
    if (sel.nrect == 0) return false;
    // segment length: one resident round of waves, segments of >= 6 rows (measured on 1024^2 blocks: 12 rows
    // beat 32 by 20-40 %, on 512^2 6 beat 12 by 35 %: the three rows of pipeline start-up per segment cost
    // less than empty wave slots)
    int strips = 0;
    for (int q = 0; q < sel.nrect; ++q) strips = strips > sel.r[q].ns ? strips : sel.r[q].ns;
    static const int seg_env = std::getenv("CMFP_MARCH_SEG") ? std::atoi(std::getenv("CMFP_MARCH_SEG")) : 0;
    int seg = seg_env;
    if (seg <= 0) {
        const int rounds = wave_slots() / (strips * mult);  // segments per strip that fit one round
        seg = rounds > 0 ? (n1 + rounds - 1) / rounds : n1;
        // three rounds of shorter segments balance the tail better where they stay long (4096^2: 48 rows
        // against 142, hydro launch 3 % faster; 2048^2: no difference between 12 and 36); two states in one
        // launch keep the segments of one (4096^2: 48 rows, six rounds)
        const int rounds1 = wave_slots() / strips;
        const int seg3 = rounds1 > 0 ? (n1 + 3 * rounds1 - 1) / (3 * rounds1) : n1;
        if (seg3 >= 36) seg = seg3;
        static const int seg_min = std::getenv("CMFP_MARCH_SEG_MIN") ? std::atoi(std::getenv("CMFP_MARCH_SEG_MIN")) : 6;
        if (seg < seg_min) seg = seg_min;
        seg += seg & 1;  // the kernel runs two rows per loop trip over seg + 2 rows
    }
    sel.seg = seg;
    sel.wrap = 0;
    sel.nwaves = 0;
    for (int q = 0; q < sel.nrect; ++q) {
        MarchRect& r = sel.r[q];
        r.nseg = (r.jhi - r.jlo + seg) / seg;
        r.first = sel.nwaves;
        sel.nwaves += r.ns * r.nseg;
    }
    return true;
}

}  // namespace CMFP_NS
