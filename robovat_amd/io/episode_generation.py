"""Episode loop with the contract of ``robovat/io/episode_generation.py:21-119``
(transition dicts of state / action / reward / info; an exception discards the
episode and the loop continues).  The wall-clock SIGALRM timeout of the
reference is replaced by the on-device phase budgets, which bound every step."""
import socket
import time
import traceback


def generate_episode(env, policy, num_steps=None, debug=False):
    t = 0
    transitions = []
    observation = env.reset()
    while True:
        action = policy.action(observation)
        new_observation, reward, done, info = env.step(action)
        transitions.append({'state': observation, 'action': action, 'reward': reward, 'info': info})
        observation = new_observation
        if done:
            break
        t += 1
        if num_steps is not None and t >= num_steps:
            break
    return {'hostname': socket.gethostname(), 'timestamp': time.strftime('%Y-%m-%d-%H-%M-%S'),
            'transitions': transitions}


def generate_episodes(env, policy, num_steps=None, num_episodes=None, timeout=30, debug=False):
    episode_index = 0
    while num_episodes is None or episode_index < num_episodes:
        try:
            episode = generate_episode(env, policy, num_steps, debug)
            yield episode_index, episode
            episode_index += 1
        except Exception:   # the reference discards the episode and carries on
            traceback.print_exc()
