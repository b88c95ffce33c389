set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 900 python tests/tools/parity_campaign.py > gpurun_out/$T/campaign.txt 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/$T/campaign.txt | tail -25
timeout -k 10 300 python tests/tools/geo_parity.py > gpurun_out/$T/geo.txt 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/$T/geo.txt | tail -8
