"""Model sections of the reference's YAMLs as plain dicts (the values define the hot-path shapes).

C2 = configs/nerf-blender.yaml:19-67, C3 = configs/neus-blender.yaml:19-75 (file:line under /root/reference).
"""
import copy

NERF_BLENDER = dict(  # BASELINE.json configs[1]
    name="nerf", radius=1.5, num_samples_per_ray=1024, train_num_rays=256, max_train_num_rays=8192,
    grid_prune=True, dynamic_ray_sampling=True, batch_image_sampling=True, randomized=True, ray_chunk=32768,
    learned_background=False, background_color="random",
    geometry=dict(
        name="volume-density", radius=1.5, feature_dim=16, density_activation="trunc_exp", density_bias=-1,
        xyz_encoding_config=dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                 base_resolution=16, per_level_scale=1.447269237440378),
        mlp_network_config=dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                n_hidden_layers=1)),
    texture=dict(
        name="volume-radiance", input_feature_dim=16,
        dir_encoding_config=dict(otype="SphericalHarmonics", degree=4),
        mlp_network_config=dict(otype="FullyFusedMLP", activation="ReLU", output_activation="Sigmoid", n_neurons=64,
                                n_hidden_layers=2)),
)

NEUS_BLENDER = dict(  # BASELINE.json configs[2]
    name="neus", radius=1.5, num_samples_per_ray=1024, train_num_rays=256, max_train_num_rays=8192,
    grid_prune=True, grid_prune_occ_thre=0.001, dynamic_ray_sampling=True, batch_image_sampling=True, randomized=True,
    ray_chunk=4096, cos_anneal_end=20000, learned_background=False, background_color="random",
    variance=dict(init_val=0.3, modulate=False),
    geometry=dict(
        name="volume-sdf", radius=1.5, feature_dim=13, grad_type="analytic",
        xyz_encoding_config=dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                 base_resolution=32, per_level_scale=1.3195079107728942, include_xyz=True),
        mlp_network_config=dict(otype="VanillaMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                n_hidden_layers=1, sphere_init=True, sphere_init_radius=0.5, weight_norm=True)),
    texture=dict(
        name="volume-radiance", input_feature_dim=16,  # feature_dim + 3 (surface normal)
        dir_encoding_config=dict(otype="SphericalHarmonics", degree=4),
        mlp_network_config=dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                n_hidden_layers=2),
        color_activation="sigmoid"),
)

# configs/neuralangelo-dtu-wmask.yaml:35-75: progressive hash levels + finite-difference gradients (C5 shapes)
NEURALANGELO = copy.deepcopy(NEUS_BLENDER)
NEURALANGELO.update(radius=1.0, cos_anneal_end=0, grid_prune_occ_thre=0.001)
NEURALANGELO["geometry"].update(
    radius=1.0, grad_type="finite_difference", finite_difference_eps="progressive",
    xyz_encoding_config=dict(otype="ProgressiveBandHashGrid", n_levels=16, n_features_per_level=2,
                             log2_hashmap_size=19, base_resolution=32, per_level_scale=1.3195079107728942,
                             include_xyz=True, start_level=4, start_step=0, update_steps=1000))


# configs/neus-dtu.yaml:12-110 (C4): NeuS foreground + learned NeRF++ background, fp32 VanillaMLP heads everywhere
_VANILLA = dict(otype="VanillaMLP", activation="ReLU", output_activation="none", n_neurons=64)
NEUS_DTU = copy.deepcopy(NEUS_BLENDER)
NEUS_DTU.update(radius=1.0, ray_chunk=2048, learned_background=True, num_samples_per_ray_bg=64)
NEUS_DTU["geometry"].update(radius=1.0)
NEUS_DTU["texture"].update(mlp_network_config=dict(_VANILLA, n_hidden_layers=2))
NEUS_DTU["geometry_bg"] = dict(
    name="volume-density", radius=1.0, feature_dim=8, density_activation="trunc_exp", density_bias=-1,
    xyz_encoding_config=dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                             base_resolution=32, per_level_scale=1.3195079107728942),
    mlp_network_config=dict(_VANILLA, n_hidden_layers=1))
NEUS_DTU["texture_bg"] = dict(
    name="volume-radiance", input_feature_dim=8, dir_encoding_config=dict(otype="SphericalHarmonics", degree=4),
    mlp_network_config=dict(_VANILLA, n_hidden_layers=2), color_activation="sigmoid")


def get(name):
    return copy.deepcopy({"nerf-blender": NERF_BLENDER, "neus-blender": NEUS_BLENDER, "neuralangelo": NEURALANGELO,
                          "neus-dtu": NEUS_DTU}[name])
