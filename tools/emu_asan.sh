#!/bin/bash
# AddressSanitizer build of the CPU emulation library + the emulator tests through it (test tool; how the fixed-size
# tiles that latent 16 overran were found).  bash tools/emu_asan.sh [pytest args, default: tests/test_emu_kernels.py -x -q]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${DOF_ASAN_OUT:-/tmp/dof_emu_asan}
make -C $ROOT/deepof_amd/csrc emu EMU_OUT=$OUT -j6 \
  EMUFLAGS="-O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -DDOF_EMU -I$ROOT/include -I. -I$ROOT/tests/emu -Wno-unused-result -Wno-unknown-pragmas" \
  > /dev/null || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
cd $ROOT
if [ $# -eq 0 ]; then set -- tests/test_emu_kernels.py -x -q; fi
DOF_EMU_SO=$OUT/libdeepof_emu.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest -p no:xdist "$@"
