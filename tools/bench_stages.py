"""Print ms_per_step and the stage averages of a bench.py JSON line read from stdin (A/B runs of knobs)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ''
d = json.loads(sys.stdin.read())
r = d['roofline']
print(tag, round(d['ms_per_step'], 4), round(d.get('ms_per_step_min', 0), 4),
      {k: round(v * 1e3, 1) for k, v in r.get('stage_avg_ms', {}).items()}, 'traffic', r.get('traffic'))
