#!/bin/bash
# The CPU test suite and the loader fuzzer against an AddressSanitizer + UndefinedBehaviorSanitizer build of the library's HOST code
# (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: device code is not instrumented -- GPU ASan is not available on this pool).
# Runs in the build container (no GPU needed).  A report lands in /tmp/asan_gfx.<pid>; none = clean.
#   bash tools/asan_cpu_suite.sh [fuzz mutations per seed, default 1500]
set -eu
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, os, subprocess
sys.path.insert(0, 'gfxexp_amd')
import build as B
from concurrent.futures import ThreadPoolExecutor
odir = os.path.join(B.HERE, 'build_variants', 'obj_asan'); os.makedirs(odir, exist_ok=True)
san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-shared-libsan']
flags = [f for f in B.FLAGS if f != '-O3'] + ['-O1', '-g', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined'] + san
with ThreadPoolExecutor(max_workers=8) as ex:
    objs = [o for o in ex.map(lambda r: B._compile(r, odir, flags), B.SOURCES) if o]
os.makedirs(os.path.join(B.HERE, 'variants'), exist_ok=True)
subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(B.HERE, 'variants', 'libgfxexp_asan.so')] + san + objs)
PY
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
[ -f "$RT/libclang_rt.asan-x86_64.so" ] || RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=/tmp/asan_gfx
rm -f /tmp/asan_gfx.*
export LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so GFX_LIB=$PWD/gfxexp_amd/variants/libgfxexp_asan.so
python -m pytest tests -q -m "not gpu" -x 2>&1 | tail -3
python tools/fuzz_loaders.py "${1:-1500}"
unset LD_PRELOAD
if ls /tmp/asan_gfx.* > /dev/null 2>&1; then head -40 /tmp/asan_gfx.*; exit 1; fi
echo "sanitizers: no report"
