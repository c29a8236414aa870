cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in 16384 65536 131072; do
echo "== chunk $c"
IPC_CHUNK=$c timeout 600 python tools/ipc_zstd_read.py 2e7 4 2>&1 | grep -v "^W\|^I" | tail -5
done
