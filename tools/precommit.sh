#!/bin/bash
# Run before every commit that touches differender_amd/csrc/, include/, oracle/ or tests/: rebuilds the library and the oracle,
# then the whole CPU suite (what the driver runs: pytest -m "not gpu"). Exits non-zero on the first failure.
# (a comment tripped the oracle-isolation guard and three commits followed without a CPU run.)
set -euo pipefail
cd "$(dirname "$0")/.."
make -s -C differender_amd/csrc
make -s -C oracle
timeout 1200 python -m pytest tests/ -x -q -m "not gpu" "$@"
