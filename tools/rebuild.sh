#!/bin/bash
# rebuild everything build() builds (library, oracle, examples, tools), from any directory
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
make -C "$ROOT/verifiable-fhe-paper_amd/csrc" -j8 2>&1 | grep -E "error|warning|Error" || true
cd "$ROOT" && python -c "import __graft_entry__ as g; g.build()"
