# one-wave against four-wave front workgroups for small pipelined batches (four slots; the automatic choice only covers big batches)
for rep in 1 2 3; do for nb in 4 8 14 32 64 128; do for w in 4 1; do echo -n "wpb $w: "; timeout 120 python tools/latency_trace.py $nb 1 $w 2>&1 | tail -n 1; done; done; done
