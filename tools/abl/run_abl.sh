# usage: run_abl.sh <stem> <bench script> n1 n2 ...
cd "$GRAFT_REPO_ROOT"
STEM=$1; SCRIPT=$2; shift 2
echo base; python $SCRIPT 2>&1 | tail -1
for n in "$@"; do echo "ABL=$n"; MFP_HIP_LIB=$PWD/tools/abl/libmfp_${STEM}_$n.so python $SCRIPT 2>&1 | tail -1; done
