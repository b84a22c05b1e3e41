for q in 4 8; do for ctx in 2 3; do for mb in 256; do echo "== hwq $q ctx $ctx chunk $mb"; GPU_MAX_HW_QUEUES=$q ZK_PIPE_CTX=$ctx ZK_PIPE_CHUNK_MB=$mb python tools/e2e_probe.py 2048 2>&1 | grep -E "^decode" | tail -4; done; done; done
echo "== default env"; python tools/e2e_probe.py 2048 2>&1 | grep -E "^decode" | tail -4
