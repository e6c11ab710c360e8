"""Measurement / diagnosis scripts (none of them is imported by the product)."""


def switch_on(names):
    """Turn ON boolean composition switches by name for an A/B run: attributes of Florence2Captioner (candidate kernels, plan
    composition) or of PlanBuilder (`fuse_splitk`).  Returns the list that was set; unknown names raise."""
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.planner import PlanBuilder
    done = []
    for name in filter(None, names):
        for owner in (Florence2Captioner, PlanBuilder):
            if isinstance(vars(owner).get(name), bool):
                setattr(owner, name, True)
                done.append(name)
                break
        else:
            raise ValueError(f"unknown composition switch {name!r} (boolean class attributes of Florence2Captioner / PlanBuilder)")
    return done
