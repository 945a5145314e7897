cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 E2E_READS=200000 python tools/cli_e2e.py > /dev/null 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write spumoni_amd/bin/spumoni run -r $d/ref -p $d/sample.fa -P -c -n > /dev/null 2>&1
for i in 1 2; do SPX_TIMING=1 spumoni_amd/bin/spumoni run -r $d/ref -p $d/sample.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "build_fat|load_flat|done\." | head -12; done
rm -rf /dev/shm/e2e
