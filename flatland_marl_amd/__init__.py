from .reference_bridge import from_reference_env, dynamic_state_of_reference_env  # noqa: F401
