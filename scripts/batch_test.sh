for lg in 24 25 26 27; do
  PTGPU_BATCH_LOG2=$lg timeout -k 10 200 python3 bench.py --no-cpu --config C4 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C4 lg $lg value %.4g ms/step %.1f kernel ms %.1f' % (d['value'], d['ms_per_step'], r['launches']*r['avg_launch_ms']/d['steps']))"
done
for lg in 24 25 26; do
  PTGPU_BATCH_LOG2=$lg timeout -k 10 200 python3 bench.py --no-cpu --config C2 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 lg $lg value %.4g ms/step %.1f kernel ms %.1f' % (d['value'], d['ms_per_step'], r['launches']*r['avg_launch_ms']/d['steps']))"
done
