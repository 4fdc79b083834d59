"""Model / dataset / training configurations with the reference's JSON schema.

`hpnn()` reproduces the hyper-parameters of the reference's experiments/hpnn.json (the model the headline
benchmark is quoted on); `hpnn_neumann()` those of experiments/hpnn_neumann.json.  A user's own copy of
those JSON files loads unchanged through `load_config(path)`: strings such as "tf.nn.leaky_relu" are kept
as names and resolved by `poisson_cnn_amd.utils.convert_tf_object_names` through a lookup table instead of
the reference's eval() (utils/convert_tf_object_names.py:13-18).
"""
import copy
import json


def hpnn():
    model = {
        'use_batchnorm': True, 'use_scaling': True, 'data_format': 'channels_first', 'postsmoother_iterations': 0,
        'pre_bottleneck_convolutions_config': {
            'filters': [4, 16, 32], 'kernel_sizes': [15, 13, 11], 'padding_mode': 'symmetric',
            'activation': 'tf.nn.leaky_relu', 'use_bias': True, 'bias_initializer': 'zeros'},
        'bottleneck_deconv_config': {
            'downsampling_factors': [2, 3, 4, 8, 16], 'upsampling_factors': [2, 3, 4, 8, 16], 'filters': 32,
            'conv_kernel_sizes': [11, 9, 7, 7, 7, 5], 'deconv_kernel_sizes': [2, 3, 4, 8, 16], 'n_convs': [3, 3, 3, 3, 3, 3],
            'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.leaky_relu', 'conv_use_bias': True, 'use_resnet': True,
            'pool_downsampling_method': 'average', 'downsampling_method': 'pool'},
        'bottleneck_multilinear_config': {
            'downsampling_factors': [32, 64, 128], 'upsampling_factors': [32, 64, 128], 'filters': 32,
            'conv_kernel_sizes': [5, 5, 5], 'n_convs': [3, 3, 3], 'padding_mode': 'CONSTANT', 'constant_padding_value': 0.0,
            'conv_activation': 'tf.nn.leaky_relu', 'conv_use_bias': True, 'use_resnet': True, 'downsampling_method': 'pool',
            'pool_downsampling_method': 'average', 'resize_methods': ['bicubic', 'bilinear', 'nearest']},
        'final_convolutions_config': {
            'filters': [32, 28, 24, 20, 16, 12, 8, 4, 1], 'kernel_sizes': [15, 13, 9, 7, 5, 3, 3, 3, 3],
            'padding_mode': 'CONSTANT', 'constant_padding_value': 0.0, 'activation': 'tf.nn.leaky_relu', 'use_bias': True,
            'bias_initializer': 'zeros'},
        'scaling_config': {
            'downsampling_ratio_per_stage': 3, 'stages': 3, 'filters': 4, 'spp_levels': [[2, 2], 3, 5],
            'activation': 'tf.nn.leaky_relu', 'kernel_size': 3},
    }
    dataset = {
        'batch_size': 50, 'batches_per_epoch': 200, 'random_output_shape_range': [[192, 384], [192, 384]],
        'fourier_coeff_grid_size_range': [[1, 8], [1, 8]], 'taylor_degree_range': [[2, 6], [2, 6]],
        'grid_spacings_range': [5e-3, 5e-2], 'homogeneous_bc': True, 'return_rhses': True, 'return_boundaries': False,
        'return_dx': True, 'normalizations': {'rhs_max_magnitude': True, 'max_domain_size_squared': True},
        'uniform_grid_spacing': True,
    }
    training = {
        'n_epochs': 200, 'precision': 'float32', 'optimizer': 'adam',
        'optimizer_parameters': {'learning_rate': 1e-5, 'amsgrad': False}, 'min_learning_rate': 1e-7,
        'loss_parameters': {
            'ndims': 2, 'data_format': 'channels_first', 'mae_loss_weight': 1.0, 'integral_loss_weight': 0.4,
            'integral_loss_config': {'n_quadpts': 47, 'Lp_norm_power': 2},
            'physics_informed_loss_weight': 0.0,
            'physics_informed_loss_config': {'stencil_sizes': [5, 5], 'orders': 2, 'normalize': False},
            'scale_sample_loss_by_target_peak_magnitude': True},
    }
    return {'model': model, 'dataset': dataset, 'training': training}


def hpnn_neumann():
    cfg = hpnn()
    cfg['model']['bc_type'] = 'neumann'
    for k in ('taylor_degree_range', 'homogeneous_bc', 'return_boundaries'):
        cfg['dataset'].pop(k)
    return cfg


def hpnn_tiny():
    """A reduced-width model with the same topology, for fast CPU-side tests of host logic."""
    cfg = hpnn()
    m = cfg['model']
    m['pre_bottleneck_convolutions_config'].update(filters=[4, 8], kernel_sizes=[5, 3])
    m['bottleneck_deconv_config'].update(downsampling_factors=[2, 3], upsampling_factors=[2, 3], filters=8,
                                         conv_kernel_sizes=[5, 3], deconv_kernel_sizes=[2, 3], n_convs=[2, 2])
    m['bottleneck_multilinear_config'].update(downsampling_factors=[4, 8, 16], upsampling_factors=[4, 8, 16], filters=8,
                                              conv_kernel_sizes=[3, 3, 3], n_convs=[2, 2, 2])
    m['final_convolutions_config'].update(filters=[8, 12, 4, 4, 1], kernel_sizes=[5, 3, 3, 3, 3])
    m['scaling_config'].update(downsampling_ratio_per_stage=2, stages=2)
    return cfg


def load_config(path):
    with open(path) as f:
        return json.load(f)


def dump_config(cfg, path):
    with open(path, 'w') as f:
        json.dump(copy.deepcopy(cfg), f, indent=2)


def dbcnn():
    """experiments/dbcnn.json (Dirichlet_BC_NN_Legacy_2)."""
    model = {
        'data_format': 'channels_first', 'use_batchnorm': True,
        'boundary_conv_config': {'filters': [2, 4, 6, 8, 12, 16, 24, 27], 'kernel_sizes': [19, 17, 15, 13, 11, 9, 7, 5], 'padding_mode': 'symmetric',
                                 'activation': 'tf.nn.leaky_relu', 'use_bias': True, 'bias_initializer': 'zeros'},
        'spp_config': {'levels': [2, 3, 4, 5, 8, 11, 15, 30, 45], 'pooling_type': 'average'},
        'domain_info_mlp_config': {'units': [512, 256, 27], 'activations': ['tanh', 'tanh', 'tanh']},
        'final_convolutions_config': {'filters': [23, 19, 15, 11, 7, 5, 3, 1], 'kernel_sizes': [7, 7, 5, 5, 5, 3, 3, 3], 'padding_mode': 'CONSTANT',
                                      'constant_padding_value': 0.0, 'activation': 'tf.nn.tanh', 'use_bias': True, 'bias_initializer': 'zeros'},
        'postsmoother_iterations': 0}
    dataset = {'batch_size': 50, 'batches_per_epoch': 200, 'random_output_shape_range': [[192, 384], [192, 384]], 'random_dx_range': [5e-3, 5e-2],
               'solver_method': 'multigrid', 'boundary_random_smoothness_range': {'left': [3, 8], 'right': [3, 8], 'top': [3, 8], 'bottom': [3, 8]}}
    training = {'n_epochs': 200, 'precision': 'float32', 'optimizer': 'adam', 'optimizer_parameters': {'learning_rate': 1e-4, 'amsgrad': False},
                'min_learning_rate': 1e-7,
                'loss_parameters': {'ndims': 2, 'data_format': 'channels_first', 'mae_loss_weight': 1.0, 'integral_loss_weight': 0.4,
                                    'integral_loss_config': {'n_quadpts': 47, 'Lp_norm_power': 2}, 'physics_informed_loss_weight': 0.0,
                                    'physics_informed_loss_config': {'stencil_sizes': [5, 5], 'orders': 2, 'normalize': False},
                                    'scale_sample_loss_by_target_peak_magnitude': False}}
    return {'model': model, 'dataset': dataset, 'training': training}


def dbcnn_tiny():
    """Same structure as dbcnn() with few, narrow stages: unit tests and __graft_entry__.smoke()."""
    cfg = dbcnn()
    m = cfg['model']
    m['boundary_conv_config'].update(filters=[2, 4, 6], kernel_sizes=[7, 5, 3])
    m['spp_config']['levels'] = [2, 3, 5]
    m['domain_info_mlp_config'] = {'units': [16, 6], 'activations': ['tanh', 'tanh']}
    m['final_convolutions_config'].update(filters=[5, 3, 1], kernel_sizes=[3, 3, 3])
    cfg['training']['loss_parameters']['integral_loss_config']['n_quadpts'] = 11
    return cfg


def pcnn_end_to_end():
    """experiments/pcnn_end_to_end.json: Poisson_CNN_Legacy(hpnn, dbcnn) trained jointly on four-edge numerical samples."""
    dataset = {'batch_size': 5, 'batches_per_epoch': 200, 'random_output_shape_range': [[192, 384], [192, 384]], 'random_dx_range': [0.005, 0.05],
               'solver_method': 'multigrid', 'rhs_random_smoothness_range': [3, 8], 'randomize_rhs_smoothness': True,
               'boundary_random_smoothness_range': {'left': [3, 8], 'right': [3, 8], 'top': [3, 8], 'bottom': [3, 8]}}
    training = copy.deepcopy(dbcnn()['training'])
    training['optimizer_parameters']['learning_rate'] = 1e-5
    training['loss_parameters']['scale_sample_loss_by_target_peak_magnitude'] = True
    return {'hpnn_model': hpnn()['model'], 'dbcnn_model': dbcnn()['model'], 'dataset': dataset, 'training': training}


def pcnn_end_to_end_tiny():
    cfg = pcnn_end_to_end()
    cfg['hpnn_model'], cfg['dbcnn_model'] = hpnn_tiny()['model'], dbcnn_tiny()['model']
    cfg['dataset'].update(batch_size=2, batches_per_epoch=2, random_output_shape_range=[[40, 56], [40, 56]])
    cfg['training']['loss_parameters']['integral_loss_config']['n_quadpts'] = 11
    return cfg


def hpnn_metalearning():
    """train/hpnn_train.py layout (the model section carries model_type) for Homogeneous_Poisson_NN_Metalearning.  No experiments/*.json ships
    for that script; the hyper-parameters are those of the model file's own example (models/Homogeneous_Poisson_NN_Metalearning.py:334-377),
    training and dataset sections are hpnn.json's (the script draws reverse_poisson_dataset_generator samples, hpnn_train.py:33)."""
    dense = ['tf.nn.tanh', 'tf.nn.tanh', 'linear']
    model = {
        'model_type': 'cnn_metalearning', 'ndims': 2, 'use_batchnorm': True, 'bottleneck_upsampling': 'multilinear',
        'input_normalization': {'rhs_max_magnitude': True}, 'output_scaling': {'max_domain_size_squared': True},
        'pre_bottleneck_convolutions_config': {
            'filters': [4, 6, 8], 'kernel_sizes': [19, 17, 15], 'padding_mode': 'CONSTANT', 'conv_activation': 'tf.nn.leaky_relu',
            'dense_activations': dense, 'use_bias': False, 'bias_initializer': 'zeros', 'pre_output_dense_units': [8, 16]},
        'bottleneck_config': {
            'downsampling_factors': [1, 2, 3, 4], 'upsampling_factors': [1, 2, 3, 4], 'filters': 8, 'conv_kernel_sizes': [13, 13, 13, 13], 'n_convs': [2, 2, 2, 2],
            'conv_padding_mode': 'CONSTANT', 'conv_conv_activation': 'tf.nn.leaky_relu', 'conv_dense_activation': dense, 'conv_pre_output_dense_units': [8, 16],
            'conv_use_bias': False, 'use_resnet': True, 'conv_downsampling_kernel_sizes': [3, 2, 3, 4]},
        'final_convolutions_config': {
            'filters': [8, 6, 4, 3, 2, 1], 'kernel_sizes': [11, 7, 5, 5, 3, 3], 'padding_mode': 'CONSTANT', 'conv_activation': 'tf.nn.tanh',
            'dense_activations': dense, 'use_bias': False, 'pre_output_dense_units': [8, 16], 'bias_initializer': 'zeros', 'final_regular_conv_stages': 4},
    }
    base = hpnn()
    return {'model': model, 'dataset': base['dataset'], 'training': base['training']}


def dbcnn_metalearning_main():
    """Constructor arguments of the reference's own example of Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:219-250, run there
    with batch 10 on a 101 x 75 grid, :211-213) - the only configuration of that model the reference holds."""
    bccfg = {'filters': [4, 8, 16, 22], 'kernel_sizes': [19, 17, 15, 13, 11], 'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.tanh',
             'dense_activations': ['linear', 'linear', 'linear'], 'pre_output_dense_units': [16, 32], 'use_layernorm': True, 'use_bias': True}
    sppcfg = {'levels': [[2], 3, 5, 8], 'pooling_type': 'average'}
    mlpcfg = {'units': [250, 125, 22], 'activations': ['tf.nn.leaky_relu', 'tf.nn.leaky_relu', 'softmax']}
    fccfg = {'filters': [32, 16, 8, 4, 2, 1], 'kernel_sizes': [7, 5, 3, 3, 3, 3], 'padding_mode': 'CONSTANT', 'constant_padding_value': 0.0,
             'final_regular_conv_stages': 3, 'use_bias': True}
    return dict(ndims=2, data_format='channels_first', boundary_conv_config=bccfg, spp_config=sppcfg, domain_info_mlp_config=mlpcfg,
                final_convolutions_config=fccfg, postsmoother_iterations=0, use_batchnorm=True)


def hpnn_metalearning_tiny():
    cfg = hpnn_metalearning()
    m = cfg['model']
    m['pre_bottleneck_convolutions_config'].update(filters=[4, 6], kernel_sizes=[5, 3], pre_output_dense_units=[6, 8])
    m['bottleneck_config'].update(downsampling_factors=[1, 2, 3], upsampling_factors=[1, 2, 3], filters=5, conv_kernel_sizes=[3, 3, 3], n_convs=[2, 2, 2],
                                  conv_pre_output_dense_units=[6, 8], conv_downsampling_kernel_sizes=[3, 2, 3])
    m['final_convolutions_config'].update(filters=[6, 4, 2, 1], kernel_sizes=[3, 3, 3, 3], pre_output_dense_units=[6, 8], final_regular_conv_stages=2)
    cfg['dataset'].update(batch_size=3, batches_per_epoch=2, random_output_shape_range=[[40, 56], [40, 56]])
    cfg['training']['loss_parameters']['integral_loss_config']['n_quadpts'] = 11
    return cfg


def hpnn_plain_tiny():
    """train/hpnn_train.py with model_type 'cnn': Homogeneous_Poisson_NN (plain convolutions, the same chained-bottleneck graph)."""
    cfg = hpnn_metalearning_tiny()
    cfg['model'] = {
        'model_type': 'cnn', 'ndims': 2, 'use_batchnorm': True, 'bottleneck_upsampling': 'deconv', 'output_scaling': {'max_domain_size_squared': True},
        'pre_bottleneck_convolutions_config': {'filters': [4, 6], 'kernel_sizes': [5, 3], 'padding_mode': 'SYMMETRIC', 'activation': 'tf.nn.leaky_relu', 'use_bias': True},
        'bottleneck_config': {'downsampling_factors': [1, 2, 4], 'upsampling_factors': [1, 2, 4], 'filters': 5, 'conv_kernel_sizes': [3, 3, 3], 'n_convs': [2, 2, 2],
                              'deconv_kernel_sizes': [1, 2, 4], 'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.leaky_relu', 'conv_use_bias': True,
                              'use_resnet': True, 'downsampling_method': 'pool', 'pool_downsampling_method': 'average'},
        'final_convolutions_config': {'filters': [6, 3, 1], 'kernel_sizes': [3, 3, 3], 'padding_mode': 'CONSTANT', 'activation': 'tf.nn.tanh', 'use_bias': True},
    }
    cfg['dataset'].update(random_output_shape_range=[[40, 56], [40, 56]])
    return cfg
