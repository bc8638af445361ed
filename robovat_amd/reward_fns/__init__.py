"""Reward functions of the MI355X backend (host mirrors; the batched versions run on the device)."""
