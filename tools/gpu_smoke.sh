#!/bin/bash
# what the driver runs at the end of a round: smoke(), then the GPU suite
set -o pipefail
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 || exit 1
