# import-only stand-in (agent_chains.py imports graphviz for a plotting helper never used on the step path)
class Digraph:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("graphviz stand-in: rendering is not available")
