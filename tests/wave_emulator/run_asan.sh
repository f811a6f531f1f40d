#!/bin/sh
# tests/wave_emulator/run_asan.sh -- TEST INFRASTRUCTURE: the engine's device code (betaone_amd/csrc/*.h) under
# AddressSanitizer on the CPU (GPU sanitizers are not available on the MI355X pool).  Builds the wave-emulator
# library with -fsanitize=address and runs the emulator test suites against it.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
"$HERE/build.sh" -fsanitize=address -fno-omit-frame-pointer -o /tmp/libbetaone_emu_asan.so
cd "$ROOT"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
BO_EMU_LIB=/tmp/libbetaone_emu_asan.so python -m pytest tests/test_engine_emu.py tests/test_fast_mode_emu.py tests/test_hostrng.py tests/test_device_turn_emu.py -x -q "$@"
