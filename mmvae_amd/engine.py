"""Graph-captured training-step engine (placeholder until the captured path lands; the module path is used)."""


class StepEngine:
    @staticmethod
    def try_build(model):
        return None
