for c in "--config celebahq256" "--config edm32" "--dtype f16" "--dtype f32x3 --steps 1"; do
  n=$(echo $c | tr -d "-" | tr " " "_")
  python3 bench.py $c --no-cpu-baseline > gpurun_out/r05_bench_$n.json 2> gpurun_out/r05_bench_$n.err || exit 1
  tail -c 300 gpurun_out/r05_bench_$n.json | head -c 1 > /dev/null
  echo "$n done" >> gpurun_out/r05_bench_progress.log
done
