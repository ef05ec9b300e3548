"""MI355X-native distributed D4PG / D3PG / DDPG actor-learner framework."""
__version__ = "0.1.0"
