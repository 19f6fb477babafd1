#!/bin/bash
# VGPRs / SGPR spills / scratch per kernel whose name contains $1, for the working tree (tools/kernel_resources_diff.sh ldpc_totals)
cd "$(dirname "$0")/../projectultra_amd/csrc" || exit 1
make resource-usage 2>/dev/null | grep -E "Function Name|VGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill|Occupancy" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - 2>/dev/null | grep "$1" | sed 's/_ZN9ultra_hip3dev//' | cut -c1-230
