# usage: tools/ab.sh "<label>|<env assignments>" ...   -- runs kbench for each config, twice, interleaved
for round in 1 2; do
  for cfg in "$@"; do
    label=${cfg%%|*}; envs=${cfg#*|}
    echo "[$round] $label"
    env $envs timeout -k 10 100 python tools/kbench.py 2>&1 | grep welch
  done
done
