// device_types.h -- plain structs shared by the host side of the library (capi) and the kernels.
// They describe how the scene and the BVH sit in HBM.
#pragma once
#include <stdint.h>
#include "../../include/gfxexp.h"
#include "emitter_spans.h"

namespace gfx {

// 48-byte vertex record: three aligned 16-byte loads per vertex gather
// (shared::Vertex is 44 B, common/common_shared.h:1109-1114).
struct DevVertex {
    float px, py, pz, nx;
    float ny, nz, tx, ty;   // t = texCoord0Dir
    float tz, u, v, pad;
};
static_assert(sizeof(DevVertex) == 48, "DevVertex must be 48 bytes");

// shared::GeometryInstanceData (common/common_shared.h:1179-1193) with buffers replaced by offsets
// into the scene-wide pools.
struct DevGeomInst {
    uint32_t vertexOffset;    // into vertex pool
    uint32_t triangleOffset;  // into triangle pool (uint32 x 3 per triangle)
    uint32_t numVertices;
    uint32_t numTriangles;
    uint32_t materialSlot;
    uint32_t distOffset;      // emitterPrimDist weights/CDF offset in the light pools (or ~0u)
    uint32_t distCount;
    float distIntegral;
};
static_assert(sizeof(DevGeomInst) == 32, "DevGeomInst must be 32 bytes");

// shared::InstanceData (common/common_shared.h:1243-1251)
struct DevInstance {
    float transform[12];        // 3x4 row-major
    float curToPrevTransform[12];
    float normalMatrix[12];     // 3 rows padded to float4 (three aligned 16-byte loads)
    // ---- one aligned 16-byte chunk: everything sample_light needs after the instance search
    uint32_t distOffset;        // lightGeomInstDist offset in the light pools (or ~0u)
    uint32_t numGeomInsts;
    float distIntegral;
    uint32_t slotsOffset;       // geomInstSlots offset in the slot pool
    float uniformScale;
    uint32_t pad[3];
};
static_assert(sizeof(DevInstance) == 176, "DevInstance must be 176 bytes");

// Per (instance, geomInst-in-instance) entry, parallel to the instance's lightGeomInstDist slice of
// the light pools: what sample_light needs once the geometry instance has been chosen.
struct LightGeomRef {
    uint32_t recBase;           // first EmitterRec of this (instance, geomInst) or ~0u
    uint32_t distOffset;        // emitterPrimDist slice in the light pools
    uint32_t distCount;
    float distIntegral;
};
static_assert(sizeof(LightGeomRef) == 16, "LightGeomRef must be 16 bytes");

// Pre-transformed emitter triangle (one per emissive triangle per instance), 96 B = six aligned
// 16-byte loads instead of the ~30 scattered loads of triangle -> 3 vertices -> transform -> material:
// world positions are inst.transform * v.position computed ONCE with the same fp32 operations
// sampleLight performs per candidate (restir_di_shared.h:412-414); normals stay in object space
// (the normal matrix is applied to the interpolated normal, :501-502).
struct EmitterRec {          // 64 B: four aligned 16-byte gathers per light candidate
    float pA[3], pB[3], pC[3];
    float nA[3];             // object-space vertex normal; flat emitters (the three normals bit-equal) need no other
    float emittance[3];
    uint32_t flags;          // bits 0-14: emittance-texture slot of the material, or 0 (only then is the record's EmitterTexRef
                             // read); bits 15-30: index of the instance's normal matrix in DevScene::lightNormalMatrices;
                             // bit 31: nB / nC differ from nA -> EmitterRecExtra holds them
};
static_assert(sizeof(EmitterRec) == 64, "EmitterRec must be 64 bytes");
constexpr uint32_t kEmitterSmooth = 0x80000000u, kEmitterTexMask = 0x00007FFFu, kEmitterMatrixShift = 15u, kEmitterMatrixMask = 0xFFFFu;
// Parallel to the records: what only smooth emitters, the three-search fallback and the solid-angle sampler read.
struct EmitterRecExtra {
    float nB[3], nC[3];
    float twoOverLenNg;   // 2 / |cross(pB - pA, pC - pA)|: the per-triangle factor of the area density
    float primProb;       // weight / integral inside the owning geometry instance's distribution
};
static_assert(sizeof(EmitterRecExtra) == 32, "EmitterRecExtra must be 32 bytes");

// One entry per (instance, geomInst) pair in (instSlot asc, list order) enumeration: the
// "geometry" list the BVH is built over (bvh::Geometry + preTransform, common/bvh_builder.h:26-36).
struct DevFlatGeom {
    uint32_t instSlot;
    uint32_t geomInstSlot;
    uint32_t triBegin;   // first flattened triangle index
    uint32_t numTriangles;
};

// Build-time view of one flattened geometry inside a BVH subtree's triangle list (static or animated
// instances): localBegin = first triangle inside that subtree's list, globalBegin = DevFlatGeom::triBegin.
struct SubsetGeom {
    uint32_t instSlot;
    uint32_t geomInstSlot;
    uint32_t localBegin;
    uint32_t globalBegin;
};

// One texture of the scene (gfx_texture_set): texels live in one pool of 32-bit words.
struct DevTexture {
    uint32_t offset;      // first word of the texels in the pool (16-byte aligned)
    uint32_t width, height;
    uint32_t format;      // enum gfx_tex_format
};
static_assert(sizeof(DevTexture) == 16, "DevTexture must be 16 bytes");

// Texture coordinates of an emitter triangle + the emittance texture of its material (parallel to the emitter
// records; read only when the scene has an emittance texture at all).
struct EmitterTexRef {    // 32 B: two aligned 16-byte gathers
    float uvA[2], uvB[2], uvC[2];
    uint32_t texelOffset; // DevTexture::offset of the material's emittance texture: the texel loads do not wait for a descriptor fetch
    uint32_t dims;        // (width - 1) | (height - 1) << 14 | format << 28  (textures are at most 16384 texels wide / high)
};
static_assert(sizeof(EmitterTexRef) == 32, "EmitterTexRef must be 32 bytes");

// Everything the shading kernels need to reach the scene.
struct DevScene {
    const gfx_material* materials;
    const DevGeomInst* geomInsts;
    const DevInstance* insts;
    const DevVertex* vertices;
    const uint32_t* triangles;
    const uint32_t* geomInstSlotPool;
    const float* lightWeights;
    const float* lightProbs;             // weight / integral of the owning distribution, same indexing
    const float* lightCDF;
    const LightGeomRef* lightGeomRefs;   // indexed like the light pools (inst.distOffset + i)
    const EmitterRec* emitterRecs;
    const EmitterRecExtra* emitterRecExtras;   // parallel to emitterRecs
    const float* lightNormalMatrices;          // [16 * matrix index]: the distinct normal matrices of the emitter instances, one 64-byte
                                               // item each (three float4 rows + padding); EmitterRec::flags holds the index
    uint32_t numLightMatrices;
    const float* lightInstIntegral; // device float[4]: [0] integral of the level-0 distribution,
                                    // [1] guide-table scale (cells / integral), [2] guide valid (uint32)
    const uint16_t* lightInstGuide; // guide table of the level-0 distribution (see lights.hip)
    uint32_t lightInstGuideCells;
    uint32_t lightInstDistOffset;   // level-0 distribution
    uint32_t numInsts;
    // the three levels flattened into one interval table over ul (emitter_spans.h), one span per emitter record
    const EmitterSpan* spans;
    const SpanGuide* spanGuide;
    const uint32_t* spanHeader;     // device uint32[4]: [0] table usable (verified by the build), [1] records checked
    uint32_t numSpans;
    uint32_t spanGuideCells;
    // textures (texture.hip.h): descriptor table indexed by slot (slot 0 unused), texel pool, sRGB decode table
    const DevTexture* textures;
    const uint32_t* texelPool;
    const float* srgbLut;                // float[256]
    const EmitterTexRef* emitterTexRefs; // parallel to emitterRecs, or null when no material has an emittance texture
};

// ---------------------------------------------------------------- BVH8 in HBM
// 64-byte wide node (one aligned half cache line, fetched cooperatively by four lanes) + a 16-byte
// link record in a parallel array (one extra dwordx4 per visit).
//   w[0..2]   quantisation origin (fp32)
//   w[3]      ex | ey << 8 | ez << 16 | imask << 24     (scale_k = 2^(e_k - 127), 8-bit grid)
//   w[4..15]  48 plane bytes, one byte per (plane, child): lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8];
//             child s of plane p = byte (s & 3) of dword 4 + 2 p + (s >> 2), so every plane value is one
//             v_cvt_f32_ubyteN away from a float (the 6-bit packing this replaces cost a bfe + cvt pair).
//   link      { first internal child (children are contiguous, slot order),
//               first triangle record of the leaf children (contiguous, slot order), valid-slot mask, 0 }
// A leaf child holds exactly one triangle (the SAH dynamic program of the builder fills all eight
// slots before it would ever pack triangles together), so there is no per-child count.
// Child slots are assigned so that slot bit k set <=> child lies on the +k side of the node centre
// (greedy auction as in Ylitie et al. 2017), which lets traversal order children by
// (slot XOR ray-octant) without sorting.  The reference layout is the 80-byte
// CompressedInternalNode_T<8> (common/common_shared.h:756-917).
struct Bvh8Node { uint32_t w[16]; };
struct Bvh8Link { uint32_t childBase, triBase, valid, pad; };
static_assert(sizeof(Bvh8Link) == 16, "Bvh8Link must be 16 bytes");
static_assert(sizeof(Bvh8Node) == 64, "Bvh8Node must be 64 bytes");

// 64-byte triangle record: shared::TriangleStorage (common/common_shared.h:1017-1025, 48 B) plus
// the ray-independent terms of the reference's ray/triangle test (bvh_builder.cpp:1256-1258),
// precomputed by the builder with the same fp32 operations: one aligned 64-byte fetch per test,
// the same size as a node, so a wave can fetch nodes and triangles with one cooperative pattern.
struct Bvh8Tri {
    float ax, ay, az, eABx;          // pA, eAB = pB - pA
    float eABy, eABz, eCAx, eCAy;    // eCA = pA - pC
    float eCAz, nx, ny, nz;          // n = cross(eCA, eAB)
    uint32_t instSlot, geomInstSlot, primIndex;
    uint32_t flatIndex;              // position in the flattened triangle list (instance slot asc, group list order, primitive)
};
static_assert(sizeof(Bvh8Tri) == 64, "Bvh8Tri must be 64 bytes");

// 48-byte world-space triangle used inside the builder (vertices + ids).
struct BuildTri {
    float ax, ay, az, bx;
    float by, bz, cx, cy;
    float cz; uint32_t instSlot, geomInstSlot, primIndex;
};
static_assert(sizeof(BuildTri) == 48, "BuildTri must be 48 bytes");

struct DevAccel {
    const Bvh8Node* nodes;
    const Bvh8Link* links;   // per node
    const Bvh8Tri* tris;
    uint32_t numNodes;
    uint32_t numTris;
    uint32_t triItemOffset;  // tris == reinterpret_cast<const Bvh8Tri*>(nodes + triItemOffset)
};

} // namespace gfx
