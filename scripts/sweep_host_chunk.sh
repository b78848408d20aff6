for c in 262144 393216 524288 786432 1048576; do
  echo -n "chunk $c: "; RTGI_HIP_HOST_CHUNK=$c timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["host_buffer_path"]["value"], d["host_buffer_path"].get("page_locked"))'
done
