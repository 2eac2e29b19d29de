"""Host mirror of Madarch.Components (reference madarch/madarch-components.ads):
a component is a (name, kind) pair compared by identity."""


class Component:
    __slots__ = ("name", "kind")

    def __init__(self, name, kind):
        self.name = name
        self.kind = kind

    def __repr__(self):
        return "Component(%r)" % self.name


def Create(name, kind):
    return Component(name, kind)


def Get_Name(c):
    return c.name


def Get_Kind(c):
    return c.kind
