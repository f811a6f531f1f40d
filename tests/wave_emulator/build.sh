#!/bin/sh
# tests/wave_emulator/build.sh -- TEST INFRASTRUCTURE: builds the engine's device code against the
# 64-lane CPU emulator (wave_emu.h) so kernel logic can be exercised in this CPU-only container.
# The product library is betaone_amd/csrc/libbetaone_hip.so (hipcc, gfx950); this one is never
# loaded by betaone_amd/.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
g++ -std=c++17 -O2 -g -fPIC -shared -DBO_WAVE_EMU -Wall -Wno-unknown-pragmas -Wno-unused-function \
    -ffp-contract=off -fno-fast-math -I"$HERE" -I"$ROOT/betaone_amd/csrc" \
    "$ROOT/betaone_amd/csrc/bo_engine.cpp" -o "$HERE/libbetaone_emu.so" -lm "$@"
