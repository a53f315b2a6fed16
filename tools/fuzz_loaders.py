"""Mutation fuzzing of the host loaders (gfxexp_amd/csrc/host/scene_builder.cpp: EXR / PFM / PNM / BMP / TGA decoders, OBJ + MTL parser) against
the ASan + UBSan build of the library's host code (tools/asan_cpu_suite.sh builds it and runs this).  Valid files are written here (the EXR
writer of tests/test_exr_reader.py), then truncated, byte-flipped, given extreme 32-bit fields or spliced; a loader may refuse a file
(GfxError) or load it -- a sanitizer report is the failure.  usage: fuzz_loaders.py [mutations per seed file, default 400]"""
import os, sys, struct, random, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfxexp_amd import api
from tests import test_exr_reader as X
rng = random.Random(7)
nrng = np.random.default_rng(3)
tmp = tempfile.mkdtemp()
seeds = {}
h, w = 9, 13
img = nrng.random((h, w)).astype(np.float32)
for comp in (X.NONE, X.RLE, X.ZIPS, X.ZIP):
    seeds['e%d.exr' % comp] = X._exr({"R": (X.HALF, img), "G": (X.FLOAT, img * 2), "B": (X.UINT, (img * 100).astype(np.uint32)), "A": (X.HALF, img)}, comp)
seeds['big.exr'] = X._exr({"Y": (X.HALF, nrng.random((40, 300)).astype(np.float32))}, X.ZIP)
seeds['a.pfm'] = b"PF\n%d %d\n-1.0\n" % (w, h) + nrng.random((h, w, 3)).astype('<f4').tobytes()
seeds['b.pfm'] = b"Pf\n%d %d\n1.0\n" % (w, h) + nrng.random((h, w)).astype('>f4').tobytes()
seeds['a.ppm'] = b"P6\n# c\n%d %d\n255\n" % (w, h) + nrng.integers(0, 255, (h, w, 3), dtype=np.uint8).tobytes()
seeds['a.pgm'] = b"P5\n%d %d\n255\n" % (w, h) + nrng.integers(0, 255, (h, w), dtype=np.uint8).tobytes()
stride = (w * 3 + 3) & ~3
seeds['a.bmp'] = b"BM" + struct.pack("<IHHI", 54 + stride * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 0, 0, 0, 0) + bytes(stride * h)
seeds['b.bmp'] = b"BM" + struct.pack("<IHHI", 54 + 4 * w * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 32, 3, 4 * w * h, 0, 0, 0, 0) + bytes(4 * w * h)
seeds['a.tga'] = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0]) + struct.pack("<HH", w, h) + bytes([24, 0x20]) + bytes(3 * w * h)
seeds['b.tga'] = bytes([3, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]) + struct.pack("<HH", w, h) + bytes([8, 0]) + b"abc" + bytes(w * h)
obj = b"""mtllib m.mtl
v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5
vt 0 0\nvt 1 0\nvt 0 1\nvt 1 1
vn 0 0 1\nvn 0 1 0
usemtl a
f 1/1/1 2/2/1 3/3/1
f -1/-1/-1 2//2 3/3
usemtl b
f 1 2 3 4
g grp
s off
f 2/2 4/4 3/3
"""
mtl = b"""newmtl a\nKd 0.5 0.5 0.5\nKs 0.1 0.1 0.1\nNs 50\nKe 1 1 1\nmap_Kd a.ppm\nmap_Bump -bm 0.5 a.tga\nnewmtl b\nKd 1 0 0\nmap_Ke a.pfm\nmap_Ks b.bmp\n"""
for n, d in seeds.items():
    open(os.path.join(tmp, n), 'wb').write(d)
open(os.path.join(tmp, 'm.mtl'), 'wb').write(mtl)
open(os.path.join(tmp, 's.obj'), 'wb').write(obj)

def mutate(d):
    d = bytearray(d)
    k = rng.random()
    if k < 0.25 and len(d) > 4:
        d = d[:rng.randrange(1, len(d))]
    elif k < 0.7:
        for _ in range(rng.randrange(1, 6)):
            d[rng.randrange(len(d))] = rng.randrange(256)
    elif k < 0.85:
        i = rng.randrange(len(d)); d[i:i + 4] = struct.pack("<I", rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 65536, 16385, len(d)]))[:max(0, min(4, len(d) - i))]
    else:
        i = rng.randrange(len(d)); j = rng.randrange(len(d)); d[i:i] = d[j:j + rng.randrange(1, 64)]
    return bytes(d)

ok = bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for n, d in seeds.items():
    s = api.HostScene()
    assert s.load_texture(os.path.join(tmp, n)) != 0, n          # the seed itself loads
    for it in range(N):
        p = os.path.join(tmp, 'mut_' + n)
        open(p, 'wb').write(mutate(d))
        s = api.HostScene()
        try:
            s.load_texture(p); ok += 1
        except api.GfxError:
            bad += 1
print('images: loaded', ok, 'refused', bad)
ok = bad = 0
s = api.HostScene(); s.load_obj(os.path.join(tmp, 's.obj'))
for it in range(N * 3):
    which = rng.random()
    open(os.path.join(tmp, 'mut.obj'), 'wb').write(mutate(obj) if which < 0.6 else obj.replace(b"m.mtl", b"mm.mtl"))
    open(os.path.join(tmp, 'mm.mtl'), 'wb').write(mutate(mtl))
    s = api.HostScene()
    try:
        s.load_obj(os.path.join(tmp, 'mut.obj')); ok += 1
    except api.GfxError:
        bad += 1
print('obj: loaded', ok, 'refused', bad)
