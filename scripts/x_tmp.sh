for pr in 0 1; do for ns in 2 3; do
echo "xs priority $pr streams $ns: $(GTX_GROUP_XS_PRIORITY=$pr GTX_GROUP_STREAMS=$ns GTX_CHUNKS_PER_WAVE=16 python scripts/share_timing.py 8 100000000 2>&1 | grep '^member 0' | cut -d, -f2-)"
done; done
