#!/usr/bin/env python3
"""end-to-end CartPole learning, reproducibly.  Runs examples/train_cartpole.py (event-ordered, HIP learner) for
several seeds plus one seed twice, and reports per seed the evaluation episode lengths after `--train-steps` updates, the first
report whose finished self-play episodes average >= 475 steps ("steps to 500"), and whether the repeated seed gave the identical run.
    python tools/learning_seeds.py --seeds 1,2,3,4,5 --train-steps 15000 --out profiles/round4/learning_cartpole_seeds.json"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(seed, steps, extra):
    with tempfile.NamedTemporaryFile(suffix='.json', delete=False) as f:
        out = f.name
    cmd = [sys.executable, os.path.join(REPO, 'examples', 'train_cartpole.py'), '--train-steps', str(steps), '--seed', str(seed), '--report-every', '500',
           '--eval-episodes', '5', '--eval-every', '1000', '--out', out] + extra
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    r = json.load(open(out))
    os.unlink(out)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='1,2,3,4,5')
    ap.add_argument('--train-steps', type=int, default=15000)
    ap.add_argument('--out', default='')
    ap.add_argument('extra', nargs='*')
    args = ap.parse_args()
    seeds = [int(s) for s in args.seeds.split(',')]
    rows, logs = [], {}
    for s in seeds:
        r = run(s, args.train_steps, args.extra)
        logs[s] = r
        to500 = next((x['train_steps'] for x in r['log'] if x['mean_episode_length'] and x['mean_episode_length'] >= 475), None)
        ev = [(x['train_steps'], sum(x['eval_episode_lengths']) / len(x['eval_episode_lengths'])) for x in r.get('evals', [])]
        first500 = next((st for st, m in ev if m >= 475), None)
        rows.append(dict(seed=s, eval_episode_lengths=r['eval_episode_lengths'], eval_mean=sum(r['eval_episode_lengths']) / len(r['eval_episode_lengths']),
                         train_steps_to_selfplay_mean_475=to500, train_steps_to_eval_mean_475=first500, eval_means_every_1000=ev,
                         evals_at_or_above_475_from_10k=sum(1 for st, m in ev if st >= 10000 and m >= 475), evals_from_10k=sum(1 for st, m in ev if st >= 10000), final_selfplay_mean=r['log'][-1]['mean_episode_length'], env_steps=r['log'][-1]['env_steps'],
                         seconds=r['log'][-1]['seconds'], final_loss=r['log'][-1]['loss']))
        print(json.dumps(rows[-1]), flush=True)
    again = run(seeds[0], args.train_steps, args.extra)
    identical = again['eval_episode_lengths'] == logs[seeds[0]]['eval_episode_lengths'] and \
        [(x['loss'], x['env_steps'], x['mean_episode_length']) for x in again['log']] == [(x['loss'], x['env_steps'], x['mean_episode_length']) for x in logs[seeds[0]]['log']]
    means = sorted(r['eval_mean'] for r in rows)
    summary = dict(train_steps=args.train_steps, seeds=seeds, eval_mean_median=means[len(means) // 2], eval_mean_min=means[0], eval_mean_max=means[-1],
                   repeated_seed=seeds[0], repeated_seed_identical=identical, rows=rows, curve_seed_first=[(x['train_steps'], x['mean_episode_length']) for x in logs[seeds[0]]['log']])
    print(json.dumps(summary))
    if args.out:
        json.dump(summary, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
