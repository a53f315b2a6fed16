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

struct RestirArgs {
    DevScene scene;
    gfx_restir_static_params s;
    gfx_restir_frame_params f;
    uint32_t curRes, baseIdx;
    size_t pixelBegin, pixelEnd;   // the launch covers pixels [pixelBegin, pixelEnd) (whole rows)
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
GFX_DEV void make_shading_point(const RestirArgs& a, uint32_t bufIdx, size_t p, f3 camPos, bool normalizeFirst, ShadingPoint& sp) {
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
