for v in "" "P2C_FPS_PPT=16" "P2C_FPS_V1=1"; do
echo "== $v"
env $v python -m pytest tests/test_gpu_parity.py -q -x -k "fps" 2>&1 | tail -1
env $v python tools/bench_sa1_forward.py --steps 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k]['ms'] for k in ('graph_serial','graph_pipelined','geometry_only')}, d['kernels']['p2c_fps_f32'])"
done
