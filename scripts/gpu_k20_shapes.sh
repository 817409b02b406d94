for a in "--clips-per-launch 32 --streams 3" "--clips-per-launch 10 --streams 2" "--clips-per-launch 7 --streams 3" "--clips-per-launch 5 --streams 4" "--clips-per-launch 10 --streams 3"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed --no-cli --legs= --sat-tiles 0 --no-parity-check $a > gpurun_out/k20s.json 2>/dev/null
  python - "$a" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/k20s.json').read().strip().splitlines()[-1])
print("%-40s %.5f ms/step %.2f M whole %.3f groups %s" % (sys.argv[1], d['ms_per_step'], d['value']/1e6, d['whole_path_frac_of_f32_peak'], d['config']['launch_groups_per_round']))
PY
done
