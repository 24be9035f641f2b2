"""Host-side naming of observation positions in the replay harness (no GPU needed)."""


def test_positions_map_to_the_feature_list_of_the_reference():
    # overcooked_features.py:38-78: the order of the leaf features inside an agent's block
    from marl_amd.replay import feature_at

    assert feature_at("global_obs", 2, 0) == "agent_dir[0] in the own block"
    assert feature_at("global_obs", 2, 4) == "inventory[0] in the own block"
    assert feature_at("global_obs", 2, 11 + 3) == "next_to_pot[3] in the own block"
    assert feature_at("global_obs", 2, 71) == "pot_0[0] in the own block"
    assert feature_at("global_obs", 2, 93) == "dist_to_other_agents[0] in the own block"
    assert feature_at("global_obs", 2, 100) == "can_move_direction[3] in the own block"
    assert feature_at("global_obs", 2, 101) == "agent_dir[0] in the block of other agent #1"
    assert feature_at("global_obs", 4, 105 * 3 + 104) == "can_move_direction[3] in the block of other agent #3"
    assert feature_at("Minimal_spatial", 2, 5) == "position 5"
