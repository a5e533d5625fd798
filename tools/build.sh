#!/bin/bash
# builds both libraries (release + range-checked); prints errors only; exits non-zero when either build fails
cd "$(dirname "$0")/../tch-geometric_amd" || exit 1
make -j8 > /tmp/tg_build.log 2>&1 || { grep -i -A6 "error" /tmp/tg_build.log | head -40; exit 1; }
make -j8 dbg > /tmp/tg_build_dbg.log 2>&1 || { grep -i -A6 "error" /tmp/tg_build_dbg.log | head -40; exit 1; }
ls -la --time-style=full-iso lib | tail -2
