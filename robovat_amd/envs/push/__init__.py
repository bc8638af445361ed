"""PushEnv: host classes around the device phase machine."""
