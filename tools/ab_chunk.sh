for c in 262144 66000 44000 33000 16500; do echo -n "BWD_CHUNK_PAIRS=$c: "; PENEO_BWD_CHUNK_PAIRS=$c timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-85; done
