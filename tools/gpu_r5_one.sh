cd /root/repo
for cfg in "ASR_NN_ROWDOT=1 ASR_RELU_BITS=1" "ASR_NN_ROWDOT=0 ASR_RELU_BITS=1" "ASR_NN_ROWDOT=1 ASR_RELU_BITS=0"; do
  echo "== $cfg"
  env $cfg ASR_FORCE_DDP=1 ASR_DDP_ONE_GRAPH=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --soak-seconds 0 --no-exposure --batch 8 2>&1 | grep -i -E "warn|fail|error|launch_mode" | cut -c1-600 | sed 's/.*"launch_mode": "\([^"]*\)".*/launch_mode: \1/'
done
