"""String-keyed plugin registry (the reference's drop-in boundary: utils/registry/registry.py:6-43).
`REG.register("Name")` as a decorator or `REG.register("Name", obj)` as a call; lookup is dict access."""


class Registry(dict):

    def register(self, name, obj=None):
        if obj is not None:
            self._add(name, obj)
            return obj

        def deco(fn):
            self._add(name, fn)
            return fn
        return deco

    def _add(self, name, obj):
        if name in self:
            raise AssertionError("%r is already registered" % (name,))
        self[name] = obj
