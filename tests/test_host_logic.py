"""Host-side logic of the product that needs no GPU (-m "not gpu")."""
def test_sorted_uniform_is_the_order_statistics_of_iid_uniforms():
    """nsr.fused_neus.sorted_uniform_ (the random cells of a grid refresh, drawn in increasing order): sorted, inside [0, 1),
    and distributed like the sorted values of i.i.d. uniforms -- the k-th of n has mean k / (n + 1)"""
    import torch
    from nsr.fused_neus import sorted_uniform_
    torch.manual_seed(3)
    n = 20000
    u = sorted_uniform_(torch.empty(n))
    assert bool((u[1:] >= u[:-1]).all()) and float(u[0]) >= 0.0 and float(u[-1]) < 1.0
    k = torch.arange(1, n + 1, dtype=torch.float64)
    # (std of the k-th order statistic <= 0.5 / sqrt(n): 6 sigma)
    assert float((u.double() - k / (n + 1)).abs().max()) < 6 * 0.5 / n ** 0.5
    cells = (u * 4096).long().clamp_(max=4095)  # what the refresh does with them: cell indices, uniformly hit
    hist = torch.bincount(cells, minlength=4096).double()
    assert abs(float(hist.mean()) - n / 4096) < 1e-9 and float(hist.max()) < 30
    assert sorted_uniform_(torch.empty(0)).numel() == 0
