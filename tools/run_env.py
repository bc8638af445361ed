#!/usr/bin/env python
"""CLI with the arguments of the reference's ``tools/run_env.py:33-155`` that
matter for the hot path: run episodes of an env with a policy on the GPU.

    python tools/run_env.py --env PushEnv --policy HeuristicPushPolicy --num_episodes 20 --seed 0
    python tools/run_env.py --env VecPushEnv --num_envs 1024 --policy RandomPolicy --num_steps 50
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', default='PushEnv')
    ap.add_argument('--policy', default='RandomPolicy')
    ap.add_argument('--num_episodes', type=int, default=5)
    ap.add_argument('--num_steps', type=int, default=None)
    ap.add_argument('--num_envs', type=int, default=1024)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--worker_id', type=int, default=0)
    ap.add_argument('--task', default=None)
    ap.add_argument('--layout_id', type=int, default=0)
    ap.add_argument('--max_steps', type=int, default=4)
    ap.add_argument('--output_dir', default=None, help='save the episodes (HDF5 layout of the reference, run_env.py:229-247)')
    ap.add_argument('--num_episodes_per_file', type=int, default=1000)
    args = ap.parse_args()
    import numpy as np
    from robovat_amd import configs, envs, policies
    from robovat_amd.io.episode_generation import generate_episodes
    np.random.seed(args.seed)
    cfg = configs.push_env_config(TASK_NAME=args.task, LAYOUT_ID=args.layout_id, MAX_STEPS=args.max_steps)
    if args.env == 'PushEnv':
        env = envs.PushEnv(config=cfg, seed=args.seed, worker_id=args.worker_id)
        policy = getattr(policies, args.policy)(env)
        t0 = time.time()
        from robovat_amd.io import hdf5_utils
        in_file, fout = 0, None            # one store stays open per output file (run_env.py:229-247)
        for i, episode in generate_episodes(env, policy, args.num_steps, args.num_episodes):
            r = sum(t['reward'] for t in episode['transitions'])
            print('episode %d: %d steps, return %.3f, %.2f s' % (i, len(episode['transitions']), r, time.time() - t0))
            if args.output_dir:
                if in_file == 0:
                    os.makedirs(args.output_dir, exist_ok=True)
                    path = os.path.join(args.output_dir, 'episodes_%s_%06d.hdf5' % (time.strftime('%Y-%m-%d-%H-%M-%S'), i))
                    fout = hdf5_utils.open_store(path, 'w')
                hdf5_utils.append_episode(fout, episode)
                in_file = (in_file + 1) % args.num_episodes_per_file
                if in_file == 0:
                    fout.close(); fout = None
        if fout is not None:
            fout.close()
    else:
        env = envs.VecPushEnv(args.num_envs, config=cfg, seed=args.seed)
        env.reset()
        t0 = time.time()
        r, d = env.rollout(args.num_steps or 5)
        env.world.synchronize()
        st = env.stats()
        print('%d envs x %d steps: %.1f env steps/s, mean reward %.3f' % (
            args.num_envs, args.num_steps or 5, st['env_steps'] / (time.time() - t0), float(r.mean())))


if __name__ == '__main__':
    main()
