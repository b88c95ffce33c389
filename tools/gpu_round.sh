set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/geo_pmc
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O -o sq -- python3 tools/geo_profile_run.py > $O/sq.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O -o lds -- python3 tools/geo_profile_run.py > $O/lds.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 tools/geo_profile_run.py > $O/kt.log 2>&1
echo rc=$?
ls -R $O | head -30
