for rep in 1 2; do for ns in 2 3 4; do
echo "rep $rep streams $ns: $(GTX_GROUP_STREAMS=$ns python scripts/share_timing.py 8 100000000 2>&1 | grep '^member' | sed 's/ of 8.*finalize [0-9.]* ms (medians of 20, events), / /' | tr '\n' '|')"
echo "rep $rep streams $ns reversed: $(GTX_GROUP_STREAMS=$ns python scripts/share_timing.py 8 100000000 rev 2>&1 | grep '^member' | sed 's/ of 8.*finalize [0-9.]* ms (medians of 20, events), / /' | tr '\n' '|')"
done; done
