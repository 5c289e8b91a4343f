#!/bin/bash
# round 4, run t2: hipGraph replay of the equaliser step with and without the runtime's packet capture
O=gpurun_out/r04_t; mkdir -p $O
for v in 1 0; do
  echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v" >> $O/eqloop_env.jsonl
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v timeout 300 python tools/eqloop.py 2>&1 | grep -v amdgpu.ids | grep -E "r04_loop\"|optimizer_launch|step_graph|step_eager" >> $O/eqloop_env.jsonl
done
cat $O/eqloop_env.jsonl
