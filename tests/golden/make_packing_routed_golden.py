"""packing_routed.json: the REFERENCE's prepare_batch (/root/reference/src/prime_rl/trainer/batch.py:796-835) on samples that carry router-replay
payloads (`TrainingSample.routed_experts`: raw bytes, [tokens, layers, top_k]) — truncation, all-or-none bins, byte concatenation, zero padding rows.
Run in the build container (needs /root/reference); msgspec is absent there, so `msgspec.Struct` is the same minimal shim make_golden.py uses."""

import importlib.util
import json
import sys
import types
from pathlib import Path

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent


def main() -> None:
    assert REF.exists(), "run in the build container: /root/reference is required"
    ms = types.ModuleType("msgspec")

    class Struct:
        def __init_subclass__(cls, **kwargs):
            super().__init_subclass__()

        def __init__(self, *args, **kwargs):
            names = list(self.__class__.__annotations__)
            for n_, v_ in zip(names, args):
                setattr(self, n_, v_)
            for n_ in names[len(args):]:
                if n_ in kwargs:
                    setattr(self, n_, kwargs.pop(n_))
                elif hasattr(self.__class__, n_):
                    setattr(self, n_, getattr(self.__class__, n_))
                else:
                    raise TypeError(f"missing field {n_}")
            assert not kwargs, kwargs

    ms.Struct = Struct
    sys.modules["msgspec"] = ms
    for name in ("prime_rl", "prime_rl.transports", "prime_rl.transports.rollouts", "prime_rl.utils", "prime_rl.trainer"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    lg = types.ModuleType("prime_rl.utils.logger")
    lg.get_logger = lambda: types.SimpleNamespace(info=lambda *a, **k: None, debug=lambda *a, **k: None, warning=lambda *a, **k: None)
    sys.modules["prime_rl.utils.logger"] = lg

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, REF / rel)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    rtypes = load("prime_rl.transports.rollouts.types", "prime_rl/transports/rollouts/types.py")
    sys.modules["prime_rl.transports.rollouts"].TrainingSample = rtypes.TrainingSample
    sys.modules["prime_rl.transports.rollouts"].MicroBatch = rtypes.MicroBatch
    batch = load("ref_batch", "prime_rl/trainer/batch.py")
    sys.path.insert(0, str(OUT))
    import packing_routed_cases as PC

    cases = []
    for case_id, (_, seq_len, workers, pad, *_rest) in enumerate(PC.CASES):
        samples = []
        for s in PC.make_samples(case_id):
            r = s.pop("routed_experts", None)
            samples.append(rtypes.TrainingSample(routed_experts=None if r is None else rtypes.RoutedExperts(**r), **s))
        grid = batch.prepare_batch(samples, seq_len, workers, batch.build_bin_cost(None), pad)
        cases.append(dict(case_id=case_id, grid=PC.digest_grid(grid)))
    (OUT / "packing_routed.json").write_text(json.dumps(dict(cases=cases)))
    print("written", OUT / "packing_routed.json", [sum(len(r) for r in c["grid"]) for c in cases])


if __name__ == "__main__":
    main()
