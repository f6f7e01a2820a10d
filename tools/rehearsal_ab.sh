ms() { python3 -c "import json,sys; o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(o['ms_per_step'],2), o['collectives_per_step']['host_ms_per_step'] if o['collectives_per_step'] else '')" "$1" "$2"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > gpurun_out/reh_a.json 2>/dev/null && ms gpurun_out/reh_a.json "single" &&
HIAST_NO_WGRAD_STREAM=1 $B > gpurun_out/reh_b.json 2>/dev/null && ms gpurun_out/reh_b.json "single, weight gradients on the main stream" &&
$B --rehearse-dist > gpurun_out/reh_c.json 2>/dev/null && ms gpurun_out/reh_c.json "rehearsal (RCCL, one rank)" &&
HIAST_NO_ASYNC_STAT=1 $B --rehearse-dist > gpurun_out/reh_d.json 2>/dev/null && ms gpurun_out/reh_d.json "rehearsal, HIAST_NO_ASYNC_STAT=1" &&
HIAST_COMM_GROUPS=0 $B --rehearse-dist > gpurun_out/reh_e.json 2>/dev/null && ms gpurun_out/reh_e.json "rehearsal, HIAST_COMM_GROUPS=0" &&
$B --rehearse-dist --backend gloo > gpurun_out/reh_f.json 2>/dev/null && ms gpurun_out/reh_f.json "rehearsal over gloo" &&
bash tools/prof_bench.sh rehearsal_streams --rehearse-dist > /dev/null 2>&1; echo prof rc=$?
