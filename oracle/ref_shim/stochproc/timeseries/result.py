class StateSpacePath:
    pass
