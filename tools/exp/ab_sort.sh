# per-kernel times of the suffix sorter (glcBwtBatch, 1024 Zipf blocks) for library builds libglc_<tag>.so: bash tools/exp/ab_sort.sh tag...
cd /tmp; export TMPDIR=/tmp
for v in ${@:-amd}; do
rm -rf /tmp/pr
GLC_LIB=/root/repo/gpu-lossless-compression_amd/libglc_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $GRAFT_REPO_ROOT/tools/probe_bwt.py 1024 4 > /tmp/log 2>&1
echo "$v $(grep -E 'bwt batch' /tmp/log | tail -1)"
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::k_fs_(sort|ties|part)" | awk -F'|' '{printf "   %-40s %s %s %s\n", substr($2,1,40), $3, $4, $5}'
done
