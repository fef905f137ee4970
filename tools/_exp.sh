cd /root/repo
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 tests/entry_protocol_worker.py > /tmp/w.log 2>&1
grep -n "rank0\]:" /tmp/w.log | head -30 | cut -c1-400
grep "ENTRY_PROTOCOL_REPORT" /tmp/w.log | cut -c1-3000
