#!/bin/bash
# Register / scratch check of ONE kernel instantiation in seconds instead of a whole-library compile:
#   tools/isa_one.sh 'gnr::k_view1_bwd_pw<false, true>(gnr::View1BwdArgs)' [extra hipcc flags]
# compiles csrc/gnr_kernels.hip without the C ABI (no launcher instantiates the templates) plus the one explicit instantiation
# into /tmp/isa_one/one-hip-amdgcn-amd-amdhsa-gfx950.s and prints the metadata of every kernel whose name matches the template's.
set -e
SIG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa_one && cd /tmp/isa_one
cat > one.hip <<EOT
#define GNR_DEV_NO_CAPI 1
#include "$ROOT/graspnerf_amd/csrc/gnr_kernels.hip"
template __global__ void $SIG;
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -save-temps -Wno-unused-value -fno-slp-vectorize "$@" one.hip -o /dev/null 2>&1 | grep -v warning | head -20
python3 - "$SIG" <<'EOT'
import re, sys
key = re.sub(r'^gnr::', '', sys.argv[1]).split('<')[0].split('(')[0]
t = open('/tmp/isa_one/one-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', t, re.S):
    blk = m.group(0)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if key in name:
        g = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk).group(1)
        print(name, 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'sgpr', g('sgpr_count'), 'scratch', g('private_segment_fixed_size'), 'spills', g('vgpr_spill_count'), 'lds', g('group_segment_fixed_size'))
EOT
