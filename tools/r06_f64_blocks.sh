# developer (ON THE GPU BOX): fp64 wide shapes, block sizes of the two passes A/B (devlibs/lib_f64*.so, tools/build_variants.sh)
for lib in "" devlibs/lib_f64a.so devlibs/lib_f64b.so devlibs/lib_f64c.so; do
  echo "== lib=$lib"; CONV3P_HIP_LIB=$lib timeout 300 python tools/generic_time.py 2>/dev/null | grep float64
done
