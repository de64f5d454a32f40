for n in "" _w3 _w2 _w5 _nosched; do
  echo "== variant libkajo_hip$n"
  KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip$n.so python bench.py --strict --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('strict', d['value'], d['roofline']['kernel_ms_per_launch'])"
  KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip$n.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fast', d['value'], d['roofline']['kernel_ms_per_launch'])"
done
