# usage: benchx.sh LIB args...   (swap the product library for an experiment build)
L=$1; shift
cp sora_amd/lib/libsora_adsb_gpu.so /tmp/lib_keep.so
cp sora_amd/lib/$L sora_amd/lib/libsora_adsb_gpu.so
timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$L', '$*', l['ms_per_step'], l['roofline']['kernel_ms'])"
cp /tmp/lib_keep.so sora_amd/lib/libsora_adsb_gpu.so
