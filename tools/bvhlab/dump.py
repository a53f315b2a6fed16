"""bvhlab input: world-space triangles of a scene + the ray sets of one ReSTIR frame (primary rays of the bench
camera, and one RIS-selected shadow ray per hit pixel: 32 emitter candidates drawn by power, one kept with
probability proportional to Le * cos * cos / d^2 -- the distribution the frame's any-hit launches see).
CPU only (uses the oracle for the primary hits).  Diagnostic tooling, not product code.

    python tools/bvhlab/dump.py [bench|small] /tmp/bvhlab
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gfxexp_amd import api  # noqa: E402
from tests import util  # noqa: E402
from tools.bvh_quality import camera_rays  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/bvhlab"
    os.makedirs(out, exist_ok=True)
    hs = util.bench_street() if which == "bench" else util.small_street()
    w, h = 480, 270
    cam = api.make_camera(w, h, pos=(1.5, 2.2, 52.0), pitch=4.0, yaw=181.5) if which == "bench" else \
        api.make_camera(w, h, pos=(2.0, 5.0, 26.0), pitch=4.0, yaw=180.0)
    org, dirs = camera_rays(cam, w, h)
    osc = util.feed_oracle(hs)
    tris = osc.world_triangles()                      # (N, 3, 3) in flattened order
    ids = osc.tri_ids()
    hits, stats = osc.trace(3, org, dirs, want_stats=True)
    n = len(org)
    print("triangles", len(tris), "primary rays", n, "oracle SAH tree: nodes/ray", int(stats[0]) / n, "tris/ray", int(stats[1]) / n)
    # emitter triangles and their power
    mats = hs.materials()
    geoms = hs.geoms()
    emit = np.array([[m.emittance[0], m.emittance[1], m.emittance[2]] if m.hasEmittance else [0, 0, 0] for m in mats], np.float32)
    geom_mat = np.array([g[2] for g in geoms], np.int64)
    tri_emit = emit[geom_mat[ids["geomInstSlot"]]]
    e0, e1 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    nrm = np.cross(e0, e1)
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    lum = tri_emit @ np.array([0.2126, 0.7152, 0.0722], np.float32)
    wgt = (lum * area).astype(np.float64)
    em_idx = np.nonzero(wgt > 0)[0]
    p = wgt[em_idx] / wgt[em_idx].sum()
    rng = np.random.default_rng(1)
    hit = hits["triIndex"] != 0xFFFFFFFF
    hp = org[hit, :3] + dirs[hit, :3] * hits["dist"][hit, None]
    # geometric normal of the hit triangle, flipped toward the camera.  triIndex indexes the oracle's own triangle
    # storage; its ids table gives the flattened position via (inst, geom, prim) -> build a lookup
    key = (ids["instSlot"].astype(np.int64) << 40) | (ids["geomInstSlot"].astype(np.int64) << 20) | ids["primIndex"].astype(np.int64)
    order = np.argsort(key)
    otri = osc.tri_ids()
    hk = (otri["instSlot"][hits["triIndex"][hit]].astype(np.int64) << 40) | (otri["geomInstSlot"][hits["triIndex"][hit]].astype(np.int64) << 20) | \
        otri["primIndex"][hits["triIndex"][hit]].astype(np.int64)
    flat = order[np.searchsorted(key[order], hk)]
    gn = nrm[flat] / np.maximum(np.linalg.norm(nrm[flat], axis=1, keepdims=True), 1e-30)
    gn = np.where(np.sum(gn * dirs[hit, :3], 1, keepdims=True) > 0, -gn, gn)
    m = hp.shape[0]
    best_w = np.zeros(m)
    wsum = np.zeros(m)
    sel_pos = np.zeros((m, 3))
    for c in range(32):
        k = em_idx[rng.choice(len(em_idx), size=m, p=p)]
        u0, u1 = rng.random(m), rng.random(m)
        su = np.sqrt(u0)
        b0, b1 = 1 - su, u1 * su
        lp = tris[k, 0] * b0[:, None] + tris[k, 1] * b1[:, None] + tris[k, 2] * (1 - b0 - b1)[:, None]
        d = lp - hp
        d2 = np.sum(d * d, 1)
        dn = d / np.sqrt(d2)[:, None]
        ln = nrm[k] / np.maximum(np.linalg.norm(nrm[k], axis=1, keepdims=True), 1e-30)
        cl = np.maximum(-np.sum(dn * ln, 1), 0)
        cs = np.maximum(np.sum(dn * gn, 1), 0)
        pdf = (wgt[k] / wgt[em_idx].sum()) / area[k]
        wt = lum[k] * cl * cs / d2 / pdf
        wsum += wt
        take = rng.random(m) * wsum < wt
        sel_pos[take] = lp[take]
        best_w[take] = wt[take]
    ok = wsum > 0
    so = hp[ok] + 1e-3 * gn[ok]
    sd = sel_pos[ok] - so
    dist = np.linalg.norm(sd, axis=1)
    sorg = np.zeros((ok.sum(), 4), np.float32)
    sdir = np.zeros((ok.sum(), 4), np.float32)
    sorg[:, :3] = so
    sdir[:, :3] = sd / dist[:, None]
    sdir[:, 3] = 0.9999 * dist
    tris.astype(np.float32).tofile(os.path.join(out, "tris.bin"))
    np.concatenate([org, dirs], 1).astype(np.float32).tofile(os.path.join(out, "rays_closest.bin"))
    np.concatenate([sorg, sdir], 1).astype(np.float32).tofile(os.path.join(out, "rays_any.bin"))
    occ = osc.trace(1, sorg, sdir)
    print("shadow rays", len(sorg), "occluded fraction", float(np.mean(occ != 0)))


if __name__ == "__main__":
    main()
