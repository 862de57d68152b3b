"""A/B runs of bench.py on one box: every spec is one subprocess.

  python tools/ab.py [--reps N] 'label|ENV=1,ENV2=x|--bench --args' ...

Prints label, avg_launch_us of each repetition and the minimum.  Specs are run
interleaved (rep 0 of every spec, then rep 1, ...) so that clock / box drift
hits all of them alike.  `--traffic` in the args adds roofline.traffic.
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = 2
specs = []
argv = sys.argv[1:]
while argv:
  a = argv.pop(0)
  if a == '--reps':
    reps = int(argv.pop(0))
  else:
    specs.append(a.split('|'))
res = {s[0]: [] for s in specs}
traffic = {}
for rep in range(reps):
  for s in specs:
    label, envs, args = s[0], s[1] if len(s) > 1 else '', s[2] if len(s) > 2 else ''
    env = dict(os.environ)
    for kv in filter(None, envs.split(',')):
      k, v = kv.split('=', 1)
      env[k] = v
    extra = args.split()
    want_traffic = '--traffic' in extra
    if want_traffic:
      extra.remove('--traffic')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline',
           '--no-extra'] + ([] if (want_traffic and rep == 0) else ['--traffic', 'off']) + extra
    try:
      out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
      line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
      j = json.loads(line)
      res[label].append(j['roofline']['avg_launch_us'])
      if j['roofline'].get('traffic'):
        traffic[label] = j['roofline']['traffic']
    except Exception as e:  # pylint: disable=broad-except
      res[label].append(float('nan'))
      sys.stderr.write('%s: %s\n%s\n' % (label, e, out.stderr[-2000:] if 'out' in dir() else ''))
    print('%-28s %s' % (label, ' '.join('%7.2f' % v for v in res[label])), flush=True)
print('---- summary (min / all)')
for s in specs:
  v = res[s[0]]
  t = traffic.get(s[0])
  print('%-28s min %7.2f   %s%s' % (s[0], min(v), ' '.join('%7.2f' % x for x in v),
                                    ('   traffic %s' % json.dumps(t)) if t else ''))
