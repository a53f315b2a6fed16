"""Procedural scenes of the benchmarks (no dependency on tests/ or oracle/).

street scenes come from the host builder `gfxh_scene_make_street` (gfxexp_amd/csrc/host/scene_builder.cpp):
instanced facades, props, lamps (pole + emissive box head) and emissive signs."""
import os

import numpy as np

from gfxexp_amd import api



def small_street(seed=7, scale=1, textured=False, cluttered=False):
    p = api.GfxhStreetParams()
    p.textured = int(textured)
    if cluttered:
        p.numTrees, p.leavesPerTree, p.numWires, p.numRailings = 6, 300, 8, 10
    p.seed = seed
    p.groundTess = 24 * scale
    p.numBuildings = 8
    p.facadeTess = 12 * scale
    p.numProps = 30 * scale
    p.propSubdiv = 1
    p.numLamps = 24 * scale
    p.numSigns = 12 * scale
    p.extent = 30.0
    p.lampEmittance = 40.0
    p.signEmittance = 8.0
    s = api.HostScene()
    s.make_street(p)
    return s


def bench_street(seed=2024, textured=False, cluttered=False):
    """The Bistro-Exterior stand-in used by bench.py (2.55 M instanced triangles, 2 745 instances, 2 100 emitters).
    textured=True: the same geometry with albedo / smoothness / normal maps on ground, facades and crates and float
    emittance maps on the signs (gfxh_scene_make_street, `textured`).
    cluttered=True: + 70 trees of 6 000 leaf cards, 120 cables and 160 railing segments (+ ~1 M thin / tiny triangles):
    the depth complexity of Bistro's vegetation and ironwork (bench.py --cluttered, a secondary workload)."""
    p = api.GfxhStreetParams()
    p.textured = int(textured)
    if cluttered:
        p.numTrees, p.leavesPerTree, p.numWires, p.numRailings = 70, 6000, 120, 160
    p.seed = seed
    p.groundTess = 512
    p.numBuildings = 44
    p.facadeTess = 64
    p.numProps = 600
    p.propSubdiv = 3
    p.numLamps = 1500
    p.numSigns = 600
    p.extent = 60.0
    p.lampEmittance = 60.0
    p.signEmittance = 10.0
    s = api.HostScene()
    s.make_street(p)
    return s


def bunny_scene(obj_path, with_light=True, with_ground=True):
    """BASELINE config 2: bunny (scale 0.1) + rectangle light + ground quad.  `obj_path`: the bunny mesh the reference's test
    harness names (stanford_bunny_309_faces.obj); the package ships no mesh data -- tests and tools pass the fixture's path."""
    s = api.HostScene()
    g = s.load_obj(obj_path)
    s.add_instance(g, api.make_transform(scale=0.1))
    if with_ground:
        mat = s.add_material_traditional((0.7, 0.7, 0.7), (0.04, 0.04, 0.04), 0.1)
        v = np.zeros(4, api.VERTEX_DTYPE)
        v["position"] = [(-20, 0, -20), (20, 0, -20), (20, 0, 20), (-20, 0, 20)]
        v["normal"] = (0, 1, 0)
        v["texCoord0Dir"] = (1, 0, 0)
        v["texCoord"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
        geom = s.add_geom(v, [(0, 2, 1), (0, 3, 2)], mat)
        s.add_instance(s.add_group([geom]), api.make_transform())
    if with_light:
        r = s.add_rectangle(1.0, 1.0, (50, 50, 50))
        s.add_instance(r, api.make_transform(pos=(0.0, 12.0, 2.0)))
        r2 = s.add_rectangle(2.0, 1.0, (10, 20, 40))
        s.add_instance(r2, api.make_transform(pitch=-60.0, pos=(-6.0, 6.0, 6.0)))
    return s
