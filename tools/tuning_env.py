"""Forced-mode runs of tests and tools (test / measurement infrastructure, not product code).

The engine library reads no environment variable: its A/B knobs are `dr_set_option` names ("fused_stack",
"fused_tail", "blocked_accumulation", "tune.tile", "tune.stack_fl", ... - include/diffroll_amd.h).  Tests and tools
that must pin a kernel flavour for a whole process (child processes of the bit-identity tests, the soaks, a
forced-mode run of the whole suite) do it here:

    DR_TEST_TUNE="fused_stack=0,tune.tile=3202" python -m pytest tests -m gpu

`install()` applies those options to every engine the process creates, as its DEFAULTS: a test that sets an option itself
still gets what it asks for.  Two kinds are PINNED (later set_option calls on them are ignored): the process-wide `tune.*`
knobs, and `blocked_accumulation`, which the facade re-asserts from its `accumulation=` keyword at every use.
"""
import os

_forced = None


def parse(spec=None):
    spec = os.environ.get("DR_TEST_TUNE", "") if spec is None else spec
    return {k.strip(): int(v) for k, v in (item.split("=", 1) for item in spec.split(",") if item.strip())}


def forced(name, default):
    """The pinned value of option `name` in this process's environment, else `default`."""
    return parse().get(name, default)


def is_forced(name):
    return name in parse()


def env_with(env=None, **options):
    """A copy of `env` (default os.environ) whose DR_TEST_TUNE also pins `options` (keys with '.' as '__':
    tune__tile=3202)."""
    env = dict(os.environ if env is None else env)
    cur = parse(env.get("DR_TEST_TUNE", ""))
    cur.update({k.replace("__", "."): int(v) for k, v in options.items()})
    env["DR_TEST_TUNE"] = ",".join(f"{k}={v}" for k, v in cur.items())
    return env


def install():
    """Patch diffroll_amd.engine.Engine once per process; returns the pinned options."""
    global _forced
    if _forced is not None:
        return _forced
    _forced = parse()
    if not _forced:
        return _forced
    from diffroll_amd.engine import Engine
    orig_init, orig_set = Engine.__init__, Engine.set_option

    def __init__(self, *a, **k):
        orig_init(self, *a, **k)
        for name, v in _forced.items():
            orig_set(self, name, v)

    def set_option(self, name, value):
        if name in _forced and (name.startswith("tune.") or name == "blocked_accumulation"):
            return
        orig_set(self, name, value)

    Engine.__init__, Engine.set_option = __init__, set_option
    return _forced
