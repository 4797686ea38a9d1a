"""Entry-point module of the gymnasium ids (`gym_agario.AgarioEnv:AgarioEnv`, /root/reference/gym_agario/__init__.py:10)."""
from agarcl_amd.gym_agario import AgarioEnv  # noqa: F401
