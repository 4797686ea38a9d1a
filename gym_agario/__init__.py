"""`import gym_agario` as with the reference (/root/reference/gym_agario/__init__.py:9-23): registers agario-grid-v0,
agario-screen-v0 and agario-gobigger-v0 with gymnasium (when it is installed) on top of the MI355X-native engine.
The implementation lives in agarcl_amd/gym_agario.py; this package only provides the reference's import names."""
from agarcl_amd.gym_agario import register   # (the class is gym_agario.AgarioEnv.AgarioEnv, as in the reference)

registered = register()
