set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit=$?"; tail -2 $O/pytest.log
NO_CONFIG4=1 bash tools/profile_all.sh r04 monkey three_sphere cube > $O/profile_all.log 2>&1; echo "profile_all exit=$?"; grep "exit=" $O/profile_all.log | tr '\n' ' '
