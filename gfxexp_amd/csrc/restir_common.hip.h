// restir_common.hip.h -- device-side pieces shared by the original (restir.hip) and rearchitected
// (restir_rearch.hip.h) ReSTIR passes: launch arguments, reservoir planes, shading-point rebuild.
#pragma once
#include "internal.h"
#include "shading.hip.h"
#include "pass_common.hip.h"

namespace gfx {

constexpr int kBlock = 256;
constexpr uint32_t kSlotSkipped = 0xFFFFFFFEu;   // SpatialSlot.raySlot: neighbour not evaluated

struct SpatialSlot {            // per (pixel, k): k = 0 self, 1..N neighbours (unbiased MIS denominators)
    float targetDensity;        // unshadowed target of the selected sample at that pixel
    uint32_t streamLength;
    uint32_t raySlot;           // GFX_INVALID_SLOT: no ray needed
};

// ---------------------------------------------------------------- pixel <-> thread mapping
// Every per-pixel kernel asks pixel_of_thread() which pixel its thread owns.  All per-pixel state lives in
// row-major arrays indexed by p = y * width + x and no kernel communicates between pixels except through those
// arrays, so the mapping changes no result -- only which pixels share a wave, a CU and an XCD:
//   mode 0  row-major: a wave is a 64 x 1 strip of one scan line (rounds 1-2).
//   mode 1  a wave is an 8 x 8 tile, a 256-thread block a 16 x 16 square (the reference's own tiling for the
//           per-tile light subsets, per_pixel_ris.cu:44-61); blocks in scan order.  Rays of a tile enter the queue
//           together, G-buffer / reservoir reads are 8 rows x 128 B, and the radius-r neighbourhoods of the spatial
//           pass overlap inside a wave.
//   mode 2  mode 1 + XCD-aware block order: hardware block b runs on XCD b % 8, so blocks are dealt out such
//           that every XCD works through whole "supertiles" (2^sx x 2^sy blocks) -- the neighbours a spatial pass
//           gathers were mostly written into / are mostly found in that XCD's own 4-MiB L2.  Supertiles (not one
//           contiguous eighth of the image per XCD) keep the XCDs balanced when cost varies over the image.
// `slot` is the thread's index in the launch: dense per-launch arrays (primary rays, their hits) use it.
struct PixelGrid {
    uint32_t width, rowBegin, rowEnd;
    uint32_t gapBegin, gapRows;         // rows [gapBegin, gapBegin + gapRows) inside [rowBegin, rowEnd) are left out (the interior of a band whose seam rows run first); 0 rows = none
    uint32_t mode;
    uint32_t blocksX, blocksY;          // 16 x 16-pixel blocks covering the rows
    uint32_t superShiftX, superShiftY;  // log2 of the supertile size in blocks
    uint32_t supersX;
    uint32_t launchBlocks;              // grid size (host side)
    const uint32_t* order;              // optional: hardware block b works as block order[b] of the launch (a permutation: cost-ordered start, restir.hip)
};
struct PixelId { size_t p; int x, y; uint32_t slot; bool valid; };
// Pixel of thread `tid` (0..255) of 256-thread block `block` of the launch; a block index past the launch owns no pixel.
GFX_DEV PixelId pixel_of_block_thread(const PixelGrid& g, uint32_t block, uint32_t tid) {
    PixelId r;
    r.slot = block * 256u + tid;
    if (g.mode == 0) {
        r.p = static_cast<size_t>(g.rowBegin) * g.width + r.slot;
        if (g.gapRows && r.p >= static_cast<size_t>(g.gapBegin) * g.width) r.p += static_cast<size_t>(g.gapRows) * g.width;
        r.valid = r.p < static_cast<size_t>(g.rowEnd) * g.width;
        r.x = static_cast<int>(r.p % g.width); r.y = static_cast<int>(r.p / g.width);
        return r;
    }
    uint32_t bx, by;
    if (g.mode == 1) { bx = block % g.blocksX; by = block / g.blocksX; }
    else {
        const uint32_t xcd = block & 7u, i = block >> 3;
        const uint32_t sh = g.superShiftX + g.superShiftY;
        const uint32_t super = ((i >> sh) << 3) | xcd, within = i & ((1u << sh) - 1u);
        const uint32_t sx = super % g.supersX, sy = super / g.supersX;   // sy beyond the last row of supertiles: by >= blocksY below
        bx = (sx << g.superShiftX) + (within & ((1u << g.superShiftX) - 1u));
        by = (sy << g.superShiftY) + (within >> g.superShiftX);
    }
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t x = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    uint32_t y = g.rowBegin + by * 16u + (wave >> 1) * 8u + (lane >> 3);
    if (g.gapRows && y >= g.gapBegin) y += g.gapRows;
    r.valid = bx < g.blocksX && by < g.blocksY && x < g.width && y < g.rowEnd;
    r.x = static_cast<int>(x); r.y = static_cast<int>(y);
    r.p = r.valid ? static_cast<size_t>(y) * g.width + x : 0;
    return r;
}

// The block of the launch this hardware block works as.
GFX_DEV uint32_t launch_block(const PixelGrid& g) { return g.order ? g.order[blockIdx.x] : blockIdx.x; }
GFX_DEV PixelId pixel_of_thread(const PixelGrid& g) { return pixel_of_block_thread(g, launch_block(g), threadIdx.x); }

// host side: the grid of a per-pixel launch over rows [rowBegin, rowEnd) (Context::pixelMap* = the mode, internal.h)
inline PixelGrid make_pixel_grid(const Context& ctx, uint32_t width, uint32_t rowBegin, uint32_t rowEnd, uint32_t gapBegin = 0, uint32_t gapEnd = 0) {
    PixelGrid g;
    g.width = width; g.rowBegin = rowBegin; g.rowEnd = rowEnd;
    g.gapBegin = gapBegin; g.gapRows = gapEnd > gapBegin ? gapEnd - gapBegin : 0u;
    g.order = nullptr;
    g.mode = static_cast<uint32_t>(ctx.tune.pixelMap);
    const uint32_t rows = rowEnd - rowBegin - g.gapRows;
    g.blocksX = (width + 15u) / 16u; g.blocksY = (rows + 15u) / 16u;
    g.superShiftX = static_cast<uint32_t>(ctx.tune.superShiftX); g.superShiftY = static_cast<uint32_t>(ctx.tune.superShiftY);
    g.supersX = (g.blocksX + (1u << g.superShiftX) - 1u) >> g.superShiftX;
    const uint32_t supersY = (g.blocksY + (1u << g.superShiftY) - 1u) >> g.superShiftY;
    if (g.mode == 0) g.launchBlocks = static_cast<uint32_t>((static_cast<size_t>(rows) * width + 255u) / 256u);
    else if (g.mode == 1) g.launchBlocks = g.blocksX * g.blocksY;
    else g.launchBlocks = (((g.supersX * supersY + 7u) / 8u) * 8u) << (g.superShiftX + g.superShiftY);
    return g;
}

struct RestirArgs {
    DevScene scene;
    gfx_restir_static_params s;
    gfx_restir_frame_params f;
    uint32_t curRes, baseIdx;
    PixelGrid px;                  // which pixel each thread of a per-pixel launch owns (whole rows [rowBegin, rowEnd))
    float4* rayOrg; float4* rayDir;
    uint32_t* rayCount;
    uint32_t* pixelRaySlot;
    const uint32_t* occluded;
    const gfx_hit* hits;
    const Bvh8Tri* tris;
    float4* shadeScratch;
    SpatialSlot* spatialScratch;
    uint32_t* rearchSlots;         // rearchitected traceShadowRays: 7 planes of ray slots per pixel
};

// ---------------------------------------------------------------- reservoir planes
struct Reservoir {
    LightSample sample;
    float sumWeights;
    uint32_t streamLength;
    GFX_DEV void reset() {
        sample.emittance = f3(0.0f); sample.position = f3(0.0f); sample.normal = f3(0.0f); sample.atInfinity = 0;
        sumWeights = 0; streamLength = 0;
    }
    GFX_DEV bool update(const LightSample& s, float weight, float u) {   // restir_di_shared.h:118-125
        sumWeights += weight;
        const bool accepted = u < weight / sumWeights;
        if (accepted) sample = s;
        ++streamLength;
        return accepted;
    }
};
GFX_DEV Reservoir load_reservoir(const void* buf, size_t numPixels, size_t p) {
    const float4* b = static_cast<const float4*>(buf);
    const float4 a = b[p], c = b[numPixels + p], d = b[2 * numPixels + p];
    Reservoir r;
    r.sample.emittance = f3(a.x, a.y, a.z);
    r.sample.position = f3(a.w, c.x, c.y);
    r.sample.normal = f3(c.z, c.w, d.x);
    r.sample.atInfinity = f2bits(d.y) & 1u;
    r.sumWeights = d.z;
    r.streamLength = f2bits(d.w);
    return r;
}
GFX_DEV void store_reservoir(void* buf, size_t numPixels, size_t p, const Reservoir& r) {
    float4* b = static_cast<float4*>(buf);
    b[p] = make_float4(r.sample.emittance.x, r.sample.emittance.y, r.sample.emittance.z, r.sample.position.x);
    b[numPixels + p] = make_float4(r.sample.position.y, r.sample.position.z, r.sample.normal.x, r.sample.normal.y);
    b[2 * numPixels + p] = make_float4(r.sample.normal.z, bits2f(r.sample.atInfinity & 1u), r.sumWeights, bits2f(r.streamLength));
}

GFX_DEV uint32_t emit_ray(bool want, f3 org, f3 dir, float tmin, float tmax, const RestirArgs& a) {
    return queue_append(want, org, dir, tmin, tmax, a.rayOrg, a.rayDir, a.rayCount);
}
// For passes in which nearly every pixel has its one ray (the visibility ray of the selected candidate, the final shadow ray):
// no compaction -- the ray of launch slot s sits at queue entry s, a thread without a ray writes an empty interval (k_trace
// retires it at the fetch) and the trace runs over all launch slots.  A dense queue needs an atomic on its head per block or
// wave, and atomics on one address retire at ~13 ns each: 0.1 ms per full-HD pass for 8 100 blocks, more than the kernel around
// them (profiles/r03_experiments.txt).  EVERY thread of the launch must call it.
GFX_DEV uint32_t emit_ray_at_slot(const PixelId& px, bool want, f3 org, f3 dir, float tmin, float tmax, const RestirArgs& a) {
    a.rayOrg[px.slot] = make_float4(org.x, org.y, org.z, want ? tmin : 0.0f);
    a.rayDir[px.slot] = make_float4(dir.x, dir.y, dir.z, want ? tmax : -1.0f);
    return want ? px.slot : GFX_INVALID_SLOT;
}

// Shading point re-derived from the quantised G-buffer (every pass does this).
struct ShadingPoint {
    f3 pos;        // offset ray origin
    f3 vOutLocal;
    float dist;
    Frame frame;
    Bsdf bsdf;
};
// normalizeFirst = false: vOut = cam - p; frontHit from the unnormalised vector; vOut /= |vOut|
//                         (optix_restir_di_kernels.cu:41-46, 320-325)
// normalizeFirst = true : vOut = normalize(cam - p); frontHit from the unit vector (:230-232, 574-577)
// (Args: RestirArgs, or pathtrace.hip's PtArgs for the NRC tracer whose first-vertex NEE is the ReSTIR reservoir -- both carry
// the static parameters `s` and the scene)
template <typename Args>
GFX_DEV void make_shading_point(const Args& a, uint32_t bufIdx, size_t p, f3 camPos, bool normalizeFirst, ShadingPoint& sp) {
    const float4 g2 = static_cast<const float4*>(a.s.gbuffer2[bufIdx])[p];
    const uint4 g3 = static_cast<const uint4*>(a.s.gbuffer3[bufIdx])[p];
    f3 pos(g2.x, g2.y, g2.z);
    const f3 ng = decode_dir(f2bits(g2.w));
    f3 vOut = camPos - pos;
    float frontHit;
    if (normalizeFirst) {
        vOut = unit(vOut);
        sp.dist = 0;
        frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
    }
    else {
        frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
        sp.dist = len(vOut);
        vOut = vOut / sp.dist;
    }
    sp.pos = offset_ray_origin(pos, frontHit * ng);
    sp.frame = Frame(decode_dir(g3.x), decode_dir(g3.y));
    sp.vOutLocal = sp.frame.to_local(vOut);
    float tu, tv;
    decode_uv(g3.z, tu, tv);
    sp.bsdf.setup(a.scene, a.scene.materials[g3.w], tu, tv);
}

// restir_di_shared.h:747-771
GFX_DEV bool test_neighbor(const RestirArgs& a, bool testGeometry, uint32_t nbBuf, int nx, int ny, float dist, f3 normal, f3 camPos) {
    if (nx < 0 || nx >= a.s.imageSizeX || ny < 0 || ny >= a.s.imageSizeY) return false;
    const size_t np = static_cast<size_t>(ny) * a.s.imageSizeX + nx;
    const uint32_t nbInst = static_cast<const uint4*>(a.s.gbuffer0[nbBuf])[np].x;
    if (nbInst == 0xFFFFFFFFu) return false;
    if (testGeometry) {
        const float4 g2 = static_cast<const float4*>(a.s.gbuffer2[nbBuf])[np];
        const uint32_t qn = static_cast<const uint4*>(a.s.gbuffer3[nbBuf])[np].x;
        const f3 nbNormal = decode_dir(qn);
        const float nbDist = len(camPos - f3(g2.x, g2.y, g2.z));
        if (fabsf(nbDist - dist) / dist > 0.1f || dot(normal, nbNormal) < 0.9f) return false;
    }
    return true;
}

} // namespace gfx
