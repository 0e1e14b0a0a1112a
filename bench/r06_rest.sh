# the tests from the one that failed on (the suite stops at the first failure)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_full}; mkdir -p $O
( time python -m pytest -m gpu -x -q $(cat bench/.r06_rest_ids.txt | tr '\n' ' ') ) > $O/pytest_gpu2.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu2.txt
