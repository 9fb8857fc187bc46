from .reference_bridge import from_reference_env, dynamic_state_of_reference_env, static_of_env, dynamic_state_of_env  # noqa: F401
