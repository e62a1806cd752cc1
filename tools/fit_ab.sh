# phase stamps of the one-pass fitting kernel, a workgroup of the first round and one of a late round (build the libraries first:
#   python tools/fit_trace.py --build;  P2C_FIT_TRACE_WG=1000 P2C_FIT_TRACE_LIB=tools/libp2c_fit_trace_wg1000.so python tools/fit_trace.py --build)
cd "$GRAFT_REPO_ROOT"
for lib in tools/libp2c_fit_trace.so tools/libp2c_fit_trace_wg1000.so; do
  [ -f $lib ] || continue
  echo "== $lib"; P2C_FIT_TRACE_LIB=$lib python tools/fit_trace.py --hard 2>&1 | grep -v amdgpu.ids
  P2C_FIT_TRACE_LIB=$lib python tools/fit_trace.py 2>&1 | grep -v amdgpu.ids
done
