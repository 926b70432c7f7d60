"""Central name -> class registry and global key/value state (API of the reference's
antmmf/common/registry.py:30-716: `registry.register_model("univl")`, `registry.get_model_class(name)`,
`registry.register(key, obj)`, `registry.get(key, default)`), restated compactly: one generic table of
categories instead of one hand-written method pair per category."""

_CATEGORIES = (
    "task", "trainer", "builder", "model", "metric", "loss", "predictor", "sampler", "adversarial", "interpreter",
    "fusion", "representation", "colearning", "alignment", "translation", "optimizer", "scheduler", "processor", "decoder",
)


class Registry:
    mapping = {f"{c}_name_mapping": {} for c in _CATEGORIES}
    mapping["state"] = {}

    @classmethod
    def _register(cls, category, name):
        def wrap(obj):
            cls.mapping[f"{category}_name_mapping"][name] = obj
            return obj

        return wrap

    @classmethod
    def _lookup(cls, category, name):
        return cls.mapping[f"{category}_name_mapping"].get(name, None)

    @classmethod
    def register(cls, name, obj):
        """Store `obj` under a dotted key in the global state (registry.register("config", cfg))."""
        path = name.split(".")
        cur = cls.mapping["state"]
        for part in path[:-1]:
            cur = cur.setdefault(part, {})
        cur[path[-1]] = obj

    @classmethod
    def get(cls, name, default=None, no_warning=False):
        cur = cls.mapping["state"]
        for part in name.split("."):
            if not isinstance(cur, dict) or part not in cur:
                return default
            cur = cur[part]
        return cur

    @classmethod
    def unregister(cls, name):
        return cls.mapping["state"].pop(name, None)


for _c in _CATEGORIES:
    setattr(Registry, f"register_{_c}", classmethod(lambda cls, name, _c=_c: cls._register(_c, name)))
    setattr(Registry, f"get_{_c}_class", classmethod(lambda cls, name, _c=_c: cls._lookup(_c, name)))

registry = Registry()
