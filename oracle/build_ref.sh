#!/bin/bash
# Builds oracle/_ref/libroialign_ref.so = the reference's own CPU ROIAlign kernel from the sources where
# they lie under /root/reference.  Test infrastructure only: tests/golden/make_golden.py drives it to write tests/golden/roialign_*.npz.
# Nothing of the reference is written to disk: lines 1-219 of pysgg/csrc/cpu/ROIAlign_cpu.cpp (the kernel templates; the ATen wrapper
# at :221-257 uses torch 1.4's AT_DISPATCH_FLOATING_TYPES(input.type(), ...) and does not build against torch 2.10) are piped to g++
# together with the extern "C" caller in oracle/roialign_ref_caller.cpp.  Real torch / Python headers, no stand-ins.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${VETO_REFERENCE:-/root/reference}
SRC="$REF/pysgg/csrc/cpu/ROIAlign_cpu.cpp"
if [ ! -f "$SRC" ]; then echo "build_ref.sh: $SRC not present (GPU box): nothing to build" >&2; exit 0; fi
mkdir -p "$HERE/_ref"
TORCH_INC=$(python3 -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "include"))')
PY_INC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
{ sed -n '1,219p' "$SRC"; cat "$HERE/roialign_ref_caller.cpp"; } | \
  g++ -x c++ - -std=c++17 -O1 -ffp-contract=off -fPIC -shared -w \
      -I"$REF/pysgg/csrc" -I"$TORCH_INC" -I"$TORCH_INC/torch/csrc/api/include" -I"$PY_INC" \
      -o "$HERE/_ref/libroialign_ref.so"
echo "built $HERE/_ref/libroialign_ref.so"
